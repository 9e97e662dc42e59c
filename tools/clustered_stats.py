"""The sieve on bench.py's clustered corpus (4096 centres, cosine ~0.9 inside a cluster, queries near centres): ms per step and the
counters; under `rocprofv3 --kernel-trace` + tools/sieve_chain_from_trace.py the step's dispatch chain.
    python tools/clustered_stats.py [rows] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(31337)
centres = torch.randn((4096, 384), generator=g, dtype=torch.float32, device=dev); centres /= centres.norm(dim=1, keepdim=True)
rows = bench.gen_clustered_rows(torch, dev, n, 384, centres, 777)
pool = 16 * B
cq = centres[torch.randint(0, 4096, (pool,), generator=g, device=dev)]
nz = torch.randn((pool, 384), generator=g, dtype=torch.float32, device=dev)
q = cq + 0.3 * nz / nz.norm(dim=1, keepdim=True)
q = (q / q.norm(dim=1, keepdim=True)).double().contiguous()
ix = DeviceIndex.from_device_ptr(rows.data_ptr(), n, 384, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); del rows
se = ShardedSearcher(local_index=ix)
ix.profile(True)
for i in range(12): se.search(q[(i % 16) * B:(i % 16 + 1) * B], 10, "sqeuclidean_dist")
torch.cuda.synchronize(); ix.profile_read(reset=True); ix.scan_stats()
t0 = time.perf_counter()
for i in range(16): se.search(q[i * B:(i + 1) * B], 10, "sqeuclidean_dist")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
ln, ms = ix.profile_read(reset=True)
print(f"n={n} B={B} clustered: {dt*1e3:.3f} ms per step = {B/dt:.0f} QPS; bracket {ms/max(ln,1):.3f} ms", ix.scan_stats(), flush=True)
