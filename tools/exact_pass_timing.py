"""Time of the exact pass alone: a search whose k exceeds the filter's lists (every query takes the exact pass), one query,
on a float32 10M x 384 and a float16 6.25M x 1024 shard."""
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
        x[c:c + m] = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32).to(x.dtype)
    ix = DeviceIndex.from_device_ptr(x.data_ptr(), n, d, 0, stream=torch.cuda.current_stream().cuda_stream, float16=f16)
    torch.cuda.synchronize(); del x
    se = ShardedSearcher(local_index=ix)
    q = torch.randn((128, d), generator=g, device=dev, dtype=torch.float64)
    for metric in ("sqeuclidean_dist", "cosine_sim"):
        for B in (1, 4, 32, 128):
            for _ in range(2): se.search(q[:B], 65, metric)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): out = se.search(q[:B], 65, metric)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            print(f"n={n} d={d} f16={f16} {metric} B={B} k=65: {dt*1e3:.2f} ms per search ({dt*1e3/B:.2f} per query), flags {sorted(set(out[3].tolist()))}", flush=True)
    ix.close(); torch.cuda.empty_cache()
