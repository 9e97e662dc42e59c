"""The N > 1 path on CPU: world_size-2 gloo, the product's ShardedSearcher (blob layout, all-gather,
library merge) with the oracle standing in for the per-shard HIP search.  Result must equal the
single-process oracle over the whole corpus, including cross-shard exact ties."""

import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus():
    rng = np.random.default_rng(123)
    docs = rng.standard_normal((1001, 48)).astype(np.float32)
    docs /= np.linalg.norm(docs, axis=1, keepdims=True)
    docs[900] = docs[17]  # exact tie across the shard boundary: lower global row must win
    docs[400] = docs[17]
    qs = np.concatenate([docs[17][None].astype(np.float64), rng.standard_normal((5, 48))])
    return docs, qs


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher, shard_bounds
    from oracle import embeddings_index as oi

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    docs, qs = _corpus()
    lo, hi = shard_bounds(len(docs), world, rank)

    def local_search(q, k, metric):
        d_, r_, c_ = np.zeros((len(q), k)), np.zeros((len(q), k), np.int64), np.zeros(len(q), np.int32)
        for i, qi in enumerate(q):
            rows, dd = oi.find_flat(qi, docs[lo:hi], metric, k)
            c_[i] = len(rows)
            d_[i, : len(rows)] = dd
            r_[i, : len(rows)] = rows + lo
        return d_, r_, c_

    s = ShardedSearcher(local_search=local_search)
    res = {}
    for metric in ("sqeuclidean_dist", "cosine_sim", "inner_product", "euclidean_dist"):
        for k in (1, 7, 10):
            with np.errstate(invalid="ignore"):
                d_, r_, c_, _ = s.search(qs, k, metric)
            res[f"{metric}_{k}_d"] = d_.numpy().copy()
            res[f"{metric}_{k}_r"] = r_.numpy().copy()
            res[f"{metric}_{k}_c"] = c_.numpy().copy()
    if rank == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_single_process(tmp_path):
    from oracle import embeddings_index as oi

    out = str(tmp_path / "res.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    docs, qs = _corpus()
    for metric in ("sqeuclidean_dist", "cosine_sim", "inner_product", "euclidean_dist"):
        for k in (1, 7, 10):
            for i, q in enumerate(qs):
                with np.errstate(invalid="ignore"):
                    rows, dd = oi.find_flat(q, docs, metric, k)
                assert z[f"{metric}_{k}_c"][i] == len(rows)
                np.testing.assert_array_equal(z[f"{metric}_{k}_r"][i, : len(rows)], rows, err_msg=f"{metric} k={k} q={i}")
                np.testing.assert_array_equal(z[f"{metric}_{k}_d"][i, : len(rows)], dd)


def test_shard_bounds_cover_everything():
    from aidial_rag_amd.retrievers.sharded_index import shard_bounds

    for n in (0, 1, 7, 10_000_000):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
