"""Consecutive 128-query steps on ONE stream vs alternating between TWO streams (two searchers, two workspaces): the second
form lets step i's verify / select kernels run beside step i + 1's filter launches.  QPS over 32 steps each way; ids compared.
    python tools/two_stream_steps.py [rows] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
rows = bench.gen_rows(torch, dev, 0, n, 384)
ix = DeviceIndex.from_device_ptr(rows.data_ptr(), n, 384, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); del rows
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((32 * B, 384), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).double()
steps = 32
def run(nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    ses = [ShardedSearcher(local_index=ix) for _ in range(nstreams)]
    outs = [None] * steps
    keep = torch.zeros((steps, B, 10), dtype=torch.int64, device=dev)
    for rep in range(2):  # first repetition = warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            s = streams[i % nstreams]
            with torch.cuda.stream(s):
                o = ses[i % nstreams].search(q[i * B:(i + 1) * B], 10, "sqeuclidean_dist")
                keep[i].copy_(o[1])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt / steps, keep.clone()
t1, k1 = run(1)
t2, k2 = run(2)
t3, k3 = run(3)
print(f"n={n} B={B}: one stream {t1*1e3:.3f} ms/step = {B/t1:.0f} QPS; two streams {t2*1e3:.3f} ms/step = {B/t2:.0f} QPS; three {t3*1e3:.3f} ms/step = {B/t3:.0f} QPS; "
      f"ids identical: {bool((k1 == k2).all()) and bool((k1 == k3).all())}")
