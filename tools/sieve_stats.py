"""The sieve (csrc/vec_kernels_sieve.h) on the headline shard: ms per 128-query step, the HIP-event bracket of its two
filter launches, and its counters (candidates per query and launch, queries handed to the exact pass).
    python tools/sieve_stats.py [rows] [batch] [k] [metric]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
metric = sys.argv[4] if len(sys.argv) > 4 else "sqeuclidean_dist"
dev = torch.device("cuda:0")
rows = bench.gen_rows(torch, dev, 0, n, 384)
ix = DeviceIndex.from_device_ptr(rows.data_ptr(), n, 384, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); del rows
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((16 * B, 384), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).double()
se = ShardedSearcher(local_index=ix)
ix.profile(True)
for i in range(12): se.search(q[(i % 16) * B:(i % 16 + 1) * B], k, metric)
torch.cuda.synchronize(); ix.profile_read(reset=True); ix.scan_stats()
t0 = time.perf_counter()
for i in range(16): se.search(q[i * B:(i + 1) * B], k, metric)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
ln, ms = ix.profile_read(reset=True)
print(f"n={n} B={B} k={k} {metric}: {dt*1e3:.3f} ms per step = {B/dt:.0f} QPS; bracket {ms/max(ln,1):.3f} ms; hi image {n*384*2/(ms/max(ln,1)*1e-3)/1e12:.2f} TB/s", ix.scan_stats(), flush=True)
