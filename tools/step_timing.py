"""Where does a search step's wall time go?  Host enqueue time vs GPU time, profiling hook on/off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
n, d, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 384, 128, 10
x = torch.randn((n, d), device="cuda"); x /= x.norm(dim=1, keepdim=True)
ix = DeviceIndex.from_device_ptr(x.data_ptr(), n, d, 0)
del x
q = torch.randn((B, d), device="cuda", dtype=torch.float64)
o_row = torch.zeros((B, k), dtype=torch.int64, device="cuda"); o_dist = torch.zeros((B, k), dtype=torch.float64, device="cuda")
o_cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); o_flg = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def step():
    ix.search_device(q.data_ptr(), B, k, "sqeuclidean_dist", o_row.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), o_flg.data_ptr(), stream=st)
for prof in (False, True, False):
    ix.profile(prof)
    for _ in range(2): step()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(10):
        h0 = time.perf_counter(); step(); host.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"profiling={prof}: enqueue total {t_enq*1e3:.2f} ms, wall {t_all*1e3:.2f} ms for 10 steps; host per call us: {[int(h*1e6) for h in host]}")
    if prof: print("scan launches, ms:", ix.profile_read())
