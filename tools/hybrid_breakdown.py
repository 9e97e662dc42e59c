"""Where a hybrid step's time goes (BASELINE config C4's per-GPU share): vector leg, BM25 leg, device -> host, fusion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher
from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25, ShardedHybrid, fuse_batch
dev = torch.device("cuda:0")
n, d, B, k = 1_250_000, 384, 128, 7
g = torch.Generator(device=dev); g.manual_seed(555)
rows = torch.randn((n, d), generator=g, dtype=torch.float32, device=dev); rows /= rows.norm(dim=1, keepdim=True)
index = DeviceIndex.from_device_ptr(rows.data_ptr(), n, d, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); del rows
indptr, toks = bench.gen_bm25_corpus(np, torch, dev, n, 9000)
kw = ShardedBM25.build(indptr, toks, bench.BM25_VOCAB, doc_offset=0, device_index=0)
vec = ShardedSearcher(local_index=index)
hy = ShardedHybrid(vec, kw, k=k)
q = torch.randn((B, d), generator=g, dtype=torch.float32, device=dev); q = (q / q.norm(dim=1, keepdim=True)).double().contiguous()
qs = bench.bm25_queries(np, B, 778)
flat = torch.tensor(np.concatenate([np.asarray(x, np.int32) for x in qs]), dtype=torch.int32, device=dev)
ptr = torch.tensor(np.concatenate(([0], np.cumsum([len(x) for x in qs]))), dtype=torch.int32, device=dev)
def timeit(f, n=30):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"vector leg alone     {timeit(lambda: vec.search(q, k, 'sqeuclidean_dist')):8.1f} us")
print(f"BM25 leg alone       {timeit(lambda: kw.search(flat, k, ptr)):8.1f} us")
def both():
    vec.search(q, k, 'sqeuclidean_dist'); kw.search(flat, k, ptr)
print(f"both legs (GPU only) {timeit(both):8.1f} us")
_, v_rows, v_cnt, _ = vec.search(q, k, 'sqeuclidean_dist'); _, t_idx, t_cnt = kw.search(flat, k, ptr)
torch.cuda.synchronize()
print(f"4 x .cpu().numpy()   {timeit(lambda: (v_rows.cpu().numpy(), v_cnt.cpu().numpy(), t_idx.cpu().numpy(), t_cnt.cpu().numpy())):8.1f} us")
v = (v_rows.cpu().numpy(), v_cnt.cpu().numpy()); t = (t_idx.cpu().numpy(), t_cnt.cpu().numpy())
t0 = time.perf_counter()
for _ in range(50): fuse_batch([v, t], (1.0, 1.0), 60)
print(f"fuse_batch (host)    {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us")
print(f"whole step           {timeit(lambda: hy.search(q, 'sqeuclidean_dist', flat, ptr)):8.1f} us")
