"""C5 shape (d = 1024, float16 storage): time of one 128-query search on a single-GPU shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, k = 1024, 10
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.empty((n, d), dtype=torch.float16, device="cuda")
for c in range(0, n, 250_000):
    x[c:c + 250_000] = torch.randn((min(250_000, n - c), d), generator=g, device="cuda").half()
torch.cuda.synchronize()
t0 = time.perf_counter()
ix = DeviceIndex.from_device_ptr(x.data_ptr(), n, d, 0, float16=True)
torch.cuda.synchronize()
print(f"build {time.perf_counter()-t0:.2f} s, index HBM bytes {ix.hbm_bytes()/1e9:.2f} GB (source {n*d*2/1e9:.2f} GB)", flush=True)
del x
for B in (32, 128):
    q = torch.randn((B, d), device="cuda", dtype=torch.float64)
    o_row = torch.zeros((B, k), dtype=torch.int64, device="cuda"); o_dist = torch.zeros((B, k), dtype=torch.float64, device="cuda")
    o_cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); o_flg = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def step():
        ix.search_device(q.data_ptr(), B, k, "inner_product", o_row.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), o_flg.data_ptr(), stream=st)
    for _ in range(6): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"B={B}: {dt*1e3:.2f} ms/step, {B/dt:.0f} QPS, source-bytes rate {n*d*2/dt/1e12:.2f} TB/s, flags {int(o_flg.sum())}", ix.scan_stats(), flush=True)
