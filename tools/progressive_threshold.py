"""Step time of the 128-query scan on shards of several sizes (the per-GPU shares of a 10M-row index on 8 / 4 / 2 GPUs),
with and without the two-launch progressive scheme: run once per value of MIR_PROGRESSIVE_MIN_TILES (read at first use)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (1_250_000, 2_500_000, 5_000_000):
    x = torch.randn((n, 384), generator=g, device=dev, dtype=torch.float32)
    x /= x.norm(dim=1, keepdim=True)
    ix = DeviceIndex.from_device_ptr(x.data_ptr(), n, 384, 0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    q = torch.randn((4096, 384), generator=g, device=dev, dtype=torch.float64)
    se = ShardedSearcher(local_index=ix)
    for B in (128,):
        for i in range(15):
            se.search(q[i * B:(i + 1) * B], 10, "sqeuclidean_dist")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            se.search(q[(i % 30) * B:(i % 30 + 1) * B], 10, "sqeuclidean_dist")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        print(f"MIN_TILES={os.environ.get('MIR_PROGRESSIVE_MIN_TILES', '64')} n={n} B={B}: {dt * 1e3:.4f} ms per step, {B / dt:.0f} QPS", flush=True)
    ix.close(); del x
