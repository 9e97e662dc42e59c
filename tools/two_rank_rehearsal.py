"""Two ranks on ONE GPU (collectives over gloo): the sharded vector search, sharded BM25 and the hybrid through the real HIP
kernels, gathered blobs and merge kernels, checked against the oracle on the unsharded data - a cross-shard exact tie
included.  Not a pytest test (a GPU test process must not spawn GPU children on the build boxes); run as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/two_rank_rehearsal.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher, shard_bounds
from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25
from oracle import embeddings_index as oi
from oracle import bm25 as ob

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)  # the same data on every rank
n, d, k = 300_000, 384, 10
docs = rng.standard_normal((n, d)).astype(np.float32)
docs /= np.linalg.norm(docs, axis=1, keepdims=True)
lo0, hi0 = shard_bounds(n, world, 0)
docs[hi0 + 5] = docs[hi0 - 3]  # an exact tie across the shard boundary
qs = rng.standard_normal((40, d))
qs[3] = docs[hi0 - 3].astype(np.float64)
lo, hi = shard_bounds(n, world, rank)
shard = torch.from_numpy(docs[lo:hi]).to(dev)
ix = DeviceIndex.from_device_ptr(shard.data_ptr(), hi - lo, d, 0, row_offset=lo, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
se = ShardedSearcher(local_index=ix)
for metric in ("sqeuclidean_dist", "cosine_sim", "inner_product"):
    dist_, rows, cnt, flags = se.search(torch.from_numpy(qs).to(dev), k, metric)
    torch.cuda.synchronize()
    rows, dist_ = rows.cpu().numpy(), dist_.cpu().numpy()
    for i in range(len(qs)):
        wrows, wdist = oi.find_flat(qs[i], docs, metric, k)
        if metric == "cosine_sim":  # float32-level ties may swap: compare as the GPU tests do, by value
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs)
            assert np.allclose(alld[rows[i]], alld[wrows], rtol=0, atol=2e-7), (metric, i)
        else:
            assert list(rows[i]) == list(wrows), (metric, i, rows[i], wrows)
            assert np.allclose(dist_[i], wdist, rtol=0, atol=2e-7), (metric, i)
    if metric == "sqeuclidean_dist":
        assert list(rows[3, :2]) == [hi0 - 3, hi0 + 5], rows[3, :3]

# BM25 + hybrid: 20 000 documents over a 3 000-term vocabulary
nd, vocab = 20_000, 3000
lens = rng.integers(5, 60, nd)
indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
toks = np.minimum(rng.zipf(1.3, int(indptr[-1])) - 1, vocab - 1).astype(np.int32)
dlo, dhi = shard_bounds(nd, world, rank)
kw = ShardedBM25.build(indptr[dlo:dhi + 1] - indptr[dlo], toks[indptr[dlo]:indptr[dhi]], vocab, doc_offset=dlo, device_index=0)
queries = [rng.integers(0, vocab, rng.integers(1, 6)).astype(np.int32) for _ in range(24)]
flat = torch.tensor(np.concatenate(queries), dtype=torch.int32, device=dev)
ptr = torch.tensor(np.concatenate(([0], np.cumsum([len(x) for x in queries]))), dtype=torch.int32, device=dev)
score, idx, cnt = kw.search(flat, 7, ptr)
torch.cuda.synchronize()
corpus = [toks[indptr[i]:indptr[i + 1]].tolist() for i in range(nd)]
model = ob.BM25Okapi(corpus)
for i, q in enumerate(queries):
    s_all = model.get_scores(q.tolist())
    want_idx = ob.top_n_indexes(s_all, 7)
    assert list(idx[i].cpu().numpy()) == list(want_idx), (i, idx[i], want_idx)
    assert np.array_equal(score[i].cpu().numpy(), s_all[want_idx]), i
if rank == 0:
    print(f"two-rank rehearsal on one GPU: vector top-k (3 metrics, cross-shard tie) and sharded BM25 (bit-identical scores) equal the oracle", flush=True)
dist.barrier()
ix.close()
kw.model.close()
dist.destroy_process_group()
