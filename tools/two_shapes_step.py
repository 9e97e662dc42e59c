"""Step time (128 queries, k = 10, sqeuclidean) of the two 128-query scans: 10M x 384 float32 and 6.25M x 1024 float16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
for n, d, f16 in ((10_000_000, 384, False), (6_250_000, 1024, True)):
    x = torch.empty((n, d), device=dev, dtype=torch.float16 if f16 else torch.float32)
    for c in range(0, n, 500_000):
        m = min(500_000, n - c)
        t = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
        if not f16:
            t /= t.norm(dim=1, keepdim=True)
        x[c:c + m] = t.to(x.dtype)
    ix = DeviceIndex.from_device_ptr(x.data_ptr(), n, d, 0, stream=torch.cuda.current_stream().cuda_stream, float16=f16)
    torch.cuda.synchronize()
    del x
    q = torch.randn((4096, d), generator=g, device=dev, dtype=torch.float64)
    se = ShardedSearcher(local_index=ix)
    B = 128
    for i in range(15):
        se.search(q[i * B:(i + 1) * B], 10, "sqeuclidean_dist")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        out = se.search(q[i * B:(i + 1) * B], 10, "sqeuclidean_dist")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"n={n} d={d} f16={f16}: {dt * 1e3:.4f} ms per step, {B / dt:.0f} QPS, flagged {int((out[3] != 0).sum())}", flush=True)
    ix.close()
    torch.cuda.empty_cache()
