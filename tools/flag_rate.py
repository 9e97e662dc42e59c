"""How often the float16 scan's completeness check sends a query to the exact pass: random float16 rows (6.25M x 1024 by
default, not normalised) and random queries, by metric."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_250_000
d, k, B, steps = 1024, 10, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(11)
x = torch.empty((n, d), dtype=torch.float16, device=dev)
for c in range(0, n, 250_000):
    x[c:c + 250_000] = torch.randn((min(250_000, n - c), d), generator=g, device=dev).half()
ix = DeviceIndex.from_device_ptr(x.data_ptr(), n, d, 0, stream=torch.cuda.current_stream().cuda_stream, float16=True)
torch.cuda.synchronize(); del x
se = ShardedSearcher(local_index=ix)
for metric in ("sqeuclidean_dist", "inner_product", "cosine_sim"):
    q = torch.randn((B * steps, d), generator=g, device=dev, dtype=torch.float64)
    flagged = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        out = se.search(q[i * B:(i + 1) * B], k, metric)
        flagged += int((out[3] != 0).sum())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{metric}: {flagged} of {B * steps} queries took the exact pass; {dt / steps * 1e3:.2f} ms per {B}-query step on average", flush=True)
