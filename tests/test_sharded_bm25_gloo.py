"""Config C4's N > 1 logic on CPU: world_size-2 gloo, the product's ShardedBM25 / ShardedHybrid (statistics
all-reduce, the library's idf routine, blob layout, all-gather, library merge with the reversed tie-break, batch
fusion) with the oracle standing in for the per-shard HIP kernels.  Results must equal the single-process oracle
pipeline over the whole corpus: scores bit-identical, ids in order, including a zero-score tail that ties across
the shard boundary (ties go to the HIGHEST global index, bm25_retriever.py:84 upstream)."""

import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOCAB = 300


def _corpus():
    rng = np.random.default_rng(2)
    n = 1201
    lens = np.clip(np.round(rng.normal(30, 10, n)), 0, 80).astype(np.int64)
    lens[[5, 600, 601, 1200]] = 0  # empty documents, one at each side of the boundary and the very last one
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.minimum(rng.zipf(1.2, int(lens.sum())) - 1, VOCAB - 1).astype(np.int32)
    toks[toks == 250] = 251
    toks[indptr[900]] = 250          # term 250 occurs in exactly one document (900): every other score is 0
    toks[: indptr[700]][toks[: indptr[700]] == 260] = 261  # term 260 first appears in the SECOND shard
    vecs = rng.standard_normal((n, 32)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    queries = [[0, 1], [250], [7, 7, 19], [VOCAB + 5, 3], [299], [], [260, 2], [40, 41, 42, 43, 44]]
    qvecs = rng.standard_normal((len(queries), 32))
    return indptr, toks, vecs, queries, qvecs


class OracleShard:
    """Per-shard stand-in for DeviceBM25: the CSR restatement over the shard's documents, global statistics
    installed the way the product installs them."""

    def __init__(self, indptr, toks, lo, hi):
        from oracle import bm25 as ob

        self.lo = lo
        self.ip = indptr[lo : hi + 1] - indptr[lo]
        self.tk = toks[indptr[lo] : indptr[hi]]
        self.o = ob.BM25OkapiCSR(self.ip, self.tk, VOCAB) if len(self.tk) else None
        self.n = hi - lo

    def corpus_stats(self):
        df = self.o.df.copy() if self.o is not None else np.zeros(VOCAB, np.int64)
        first = np.full(VOCAB, np.iinfo(np.int64).max, np.int64)
        if len(self.tk):
            np.minimum.at(first, self.tk.astype(np.int64), np.arange(len(self.tk), dtype=np.int64))
        return df, first, int(len(self.tk)), self.n

    def set_global_stats(self, idf, avgdl, average_idf):
        self.idf, self.avgdl = idf, avgdl
        if self.o is not None:
            self.o.idf, self.o.avgdl, self.o.average_idf = idf, avgdl, average_idf

    def search(self, queries, k):
        from oracle import bm25 as ob

        b = len(queries)
        idx, sc, cnt = np.zeros((b, k), np.int64), np.zeros((b, k)), np.zeros(b, np.int32)
        for i, q in enumerate(queries):
            s = self.o.get_scores(q) if self.o is not None else np.zeros(self.n)
            top = ob.top_n_indexes(s, k)
            cnt[i] = len(top)
            idx[i, : len(top)] = top + self.lo
            sc[i, : len(top)] = s[top]
        return idx, sc, cnt


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25, ShardedHybrid, exchange_global_stats
    from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher, shard_bounds
    from oracle import embeddings_index as oi

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    indptr, toks, vecs, queries, qvecs = _corpus()
    lo, hi = shard_bounds(len(indptr) - 1, world, rank)
    shard = OracleShard(indptr, toks, lo, hi)
    idf, avgdl, avg_idf, n_g = exchange_global_stats(shard, VOCAB)
    sb = ShardedBM25(local_search=shard.search)

    def vec_search(q, k, metric):
        d_, r_, c_ = np.zeros((len(q), k)), np.zeros((len(q), k), np.int64), np.zeros(len(q), np.int32)
        for i, qi in enumerate(q):
            rows, dd = oi.find_flat(qi, vecs[lo:hi], metric, k)
            c_[i] = len(rows)
            d_[i, : len(rows)] = dd
            r_[i, : len(rows)] = rows + lo
        return d_, r_, c_

    res = {"idf": idf, "avgdl": avgdl, "avg_idf": avg_idf, "n": n_g}
    for k in (1, 7, 10):
        s_, i_, c_ = sb.search(queries, k)
        res[f"s{k}"], res[f"i{k}"], res[f"c{k}"] = s_.numpy().copy(), i_.numpy().copy(), c_.numpy().copy()
    hy = ShardedHybrid(ShardedSearcher(local_search=vec_search), sb, k=7)
    ids, scores, cnt, v, t = hy.search(qvecs, "sqeuclidean_dist", queries)
    res["h_ids"], res["h_scores"], res["h_cnt"] = ids, scores, cnt
    if rank == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_bm25_and_hybrid_equal_single_process(tmp_path):
    from oracle import bm25 as ob
    from oracle import embeddings_index as oi
    from oracle import fusion as of

    out = str(tmp_path / "res.npz")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    indptr, toks, vecs, queries, qvecs = _corpus()
    o = ob.BM25OkapiCSR(indptr, toks, VOCAB)
    # global statistics: bit-identical to the unsharded model's (term 260 first appears in shard 1; the idf average is
    # summed in first-appearance order)
    np.testing.assert_array_equal(z["idf"], o.idf)
    assert float(z["avgdl"]) == o.avgdl and float(z["avg_idf"]) == o.average_idf and int(z["n"]) == len(indptr) - 1
    for k in (1, 7, 10):
        for i, q in enumerate(queries):
            s = o.get_scores(q)
            top = ob.top_n_indexes(s, k)
            assert z[f"c{k}"][i] == len(top)
            np.testing.assert_array_equal(z[f"i{k}"][i], top, err_msg=f"k={k} q={q}")
            np.testing.assert_array_equal(z[f"s{k}"][i], s[top])
    # query [250]: one positive document (900, second shard); the zero tail is the highest global indices, which
    # cross the shard boundary only through the merge
    assert list(z["i10"][1][:3]) == [900, 1200, 1199]
    # hybrid: both legs k = 7, fused like retrieval_chain.py:239-245
    for i, q in enumerate(queries):
        sem, _ = oi.find_flat(qvecs[i], vecs, "sqeuclidean_dist", 7)
        bm = ob.top_n_indexes(o.get_scores(q), 7)
        want = of.weighted_reciprocal_rank([[(int(x), 0) for x in sem], [(int(x), 0) for x in bm]], [1.0, 1.0])
        got = [(int(x), 0) for x in z["h_ids"][i, : z["h_cnt"][i]]]
        assert got == want, (i, got, want)


def test_globally_empty_corpus_raises_on_every_rank():
    from aidial_rag_amd.retrievers.sharded_bm25 import exchange_global_stats

    class Empty:
        def corpus_stats(self):
            return np.zeros(4, np.int64), np.full(4, np.iinfo(np.int64).max, np.int64), 0, 3

        def set_global_stats(self, *a):
            raise AssertionError("must not be reached")

    with pytest.raises(ValueError, match="Text index is empty."):
        exchange_global_stats(Empty(), 4)
