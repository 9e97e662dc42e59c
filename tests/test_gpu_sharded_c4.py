"""BASELINE config C4 on the real kernels: hybrid semantic + BM25 + fusion over a SHARDED corpus
(retrieval_chain.py:193-252 upstream; sharding per SURVEY.md 8(e)).

* `ShardedHybrid` at world = 1 and - through `LoopbackCollective`, two shards of one process exchanging their blobs by
  device-to-device copies in exactly the layout `all_gather_into_tensor` produces - the `world > 1` branch of
  `ShardedSearcher` / `ShardedBM25` / `ShardedHybrid`: per-shard HIP search into a blob, gathered blobs with a non-zero
  shard stride into `mir_topk_merge_device`, host fusion.  Against the UNSHARDED oracle pipeline (`oracle.find_flat` +
  `oracle.bm25` + `oracle.fusion`) on a corpus whose two legs share ids, with an exact distance tie across the shard
  boundary and zero-score BM25 tails that cross it.
* BM25 at C4's per-GPU size (1.25M documents) and at the whole 10M documents on one GPU: float64 scores and top-k
  bit-identical to the CSR restatement on a query sample."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VOCAB = 3000
N, D, K = 40_000, 384, 7


@pytest.fixture(scope="module")
def corpus():
    rng = np.random.default_rng(404)
    rows = rng.standard_normal((N, D)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    cut = N // 2
    rows[cut + 5] = rows[cut - 3]  # an exact distance tie across the shard boundary: the lower global row wins
    lens = np.clip(np.round(rng.normal(60, 20, N)), 0, 150).astype(np.int64)
    lens[[7, cut - 1, cut, N - 1]] = 0  # empty documents at both sides of the boundary and at the very end
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.minimum(rng.zipf(1.15, int(lens.sum())) - 1, VOCAB - 1).astype(np.int32)
    toks[toks == 2500] = 2501
    toks[indptr[1234]] = 2500  # term 2500 occurs in exactly one document, in the FIRST shard: the zero tail is the last shard's
    toks[toks >= 2990] = 2989  # terms 2990.. occur nowhere
    # queries: the vector near a target chunk, the terms drawn from the same chunk's text -> both legs find the target
    assert lens[1234] > 0
    targets = [int(t) for t in rng.integers(0, N, 40) if lens[t] > 3]
    freq = np.bincount(toks, minlength=VOCAB)
    qvecs, qterms = [], []
    for t in targets:
        v = rows[t].astype(np.float64) + 0.05 * rng.standard_normal(D)
        qvecs.append(v)
        doc = np.unique(toks[indptr[t] : indptr[t + 1]])
        qterms.append([int(x) for x in doc[np.argsort(freq[doc], kind="stable")][:3]])  # the chunk's three rarest terms
    qvecs.append(rows[cut - 3].astype(np.float64)); qterms.append([2500])          # distance tie; one positive document + zero tail
    qvecs.append(rng.standard_normal(D)); qterms.append([2995, VOCAB + 9])          # no term known: all scores zero
    qvecs.append(rng.standard_normal(D)); qterms.append([])                         # empty keyword query
    qvecs.append(rng.standard_normal(D)); qterms.append([3, 3, 17])                 # frequent terms, one repeated
    return rows, indptr, toks, np.stack(qvecs), qterms, cut


def oracle_pipeline(rows, indptr, toks, qvecs, qterms, metric):
    from oracle import bm25 as ob
    from oracle import embeddings_index as oi
    from oracle import fusion as of

    model = ob.BM25OkapiCSR(indptr, toks, VOCAB)
    out = []
    for q, terms in zip(qvecs, qterms):
        sem, sem_d = oi.find_flat(q, rows, metric, K)
        scores = model.get_scores(terms)
        top = ob.top_n_indexes(scores, K)
        fused = of.weighted_reciprocal_rank([[int(x) for x in sem], [int(x) for x in top]], [1.0, 1.0])
        fscore = of.rrf_scores([[int(x) for x in sem], [int(x) for x in top]], [1.0, 1.0])
        out.append((sem, sem_d, top, scores[top], fused, [fscore[x] for x in fused]))
    return out


def device_queries(torch, qvecs, qterms):
    q = torch.from_numpy(np.ascontiguousarray(qvecs)).cuda()
    flat = torch.tensor(np.concatenate([np.asarray(t, np.int32) for t in qterms] + [np.zeros(0, np.int32)]), dtype=torch.int32, device="cuda")
    ptr = torch.tensor(np.concatenate(([0], np.cumsum([len(t) for t in qterms]))), dtype=torch.int32, device="cuda")
    return q, flat, ptr


def check(got, want, qterms, overlap_needed=True):
    ids, scores, cnt, v, t = got
    shared = 0
    for i, (sem, sem_d, top, top_s, fused, fscore) in enumerate(want):
        assert list(v[0][i, : v[1][i]]) == list(sem), ("vector leg", i)
        assert list(t[0][i, : t[1][i]]) == list(top), ("bm25 leg", i, qterms[i])
        assert list(ids[i, : cnt[i]]) == fused, ("fused", i)
        np.testing.assert_array_equal(scores[i, : cnt[i]], np.asarray(fscore))
        shared += len(set(int(x) for x in sem) & set(int(x) for x in top))
    if overlap_needed:
        assert shared >= len(want) // 2  # the score-summing / de-duplication path of the fusion really ran


@pytest.mark.parametrize("metric", ["sqeuclidean_dist", "inner_product"])
def test_sharded_hybrid_one_rank_equals_oracle_pipeline(corpus, metric):
    import torch

    from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
    from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
    from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25, ShardedHybrid
    from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher

    rows, indptr, toks, qvecs, qterms, cut = corpus
    ix = DeviceIndex.from_host(rows)
    kw = DeviceBM25.from_token_ids(indptr, toks, VOCAB)
    hy = ShardedHybrid(ShardedSearcher(local_index=ix), ShardedBM25(local_model=kw), k=K)
    q, flat, ptr = device_queries(torch, qvecs, qterms)
    got = hy.search(q, metric, flat, ptr)
    check(got, oracle_pipeline(rows, indptr, toks, qvecs, qterms, metric), qterms)
    ix.close()
    kw.close()


@pytest.mark.parametrize("cuts", [(0, N // 2, N), (0, 8192, 30_001, N),
                                  (0, 4_000, 9_999, 15_000, N // 2, 24_000, 31_000, 36_000, N)])  # 8-way, uneven: BASELINE C4 / C5's shard count
def test_world_gt1_branch_shards_of_one_gpu_equal_unsharded_oracle(corpus, cuts):
    """Two / three / eight shards on ONE GPU, each with its own ShardedHybrid, exchanging blobs through LoopbackCollective."""
    import torch

    from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
    from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
    from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25, ShardedHybrid, install_combined_stats
    from aidial_rag_amd.retrievers.sharded_index import LoopbackCollective, ShardedSearcher

    rows, indptr, toks, qvecs, qterms, cut = corpus
    world = len(cuts) - 1
    bus_v, bus_t = LoopbackCollective(world), LoopbackCollective(world)
    indexes, models = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        indexes.append(DeviceIndex.from_host(rows[lo:hi], row_offset=lo))
        models.append(DeviceBM25.from_token_ids(indptr[lo : hi + 1] - indptr[lo], toks[indptr[lo] : indptr[hi]], VOCAB,
                                                idf=np.zeros(VOCAB), avgdl=1.0, doc_offset=lo))
    install_combined_stats(models)
    hybrids = [ShardedHybrid(ShardedSearcher(local_index=indexes[r], collective=bus_v.for_rank(r)),
                             ShardedBM25(local_model=models[r], collective=bus_t.for_rank(r)), k=K) for r in range(world)]
    assert all(h.vector.world == world and h.keywords.world == world for h in hybrids)
    q, flat, ptr = device_queries(torch, qvecs, qterms)
    metric = "sqeuclidean_dist"
    got, errors = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            got[r] = hybrids[r].search(q, metric, flat, ptr)
        except BaseException as e:  # noqa: BLE001 - reported below, from the main thread
            errors.append((r, e))
            bus_v._barrier.abort()
            bus_t._barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    want = oracle_pipeline(rows, indptr, toks, qvecs, qterms, metric)
    for r in range(world):  # every rank holds the merged lists and fuses them itself
        check(got[r], want, qterms)
    i_tie = len(qterms) - 4
    assert list(got[0][3][0][i_tie, :2]) == [cut - 3, cut + 5]                      # cross-shard distance tie
    assert list(got[0][4][0][i_tie]) == [1234] + list(range(N - 1, N - K, -1))      # one positive (shard 0), zero tail from the last shard
    assert list(got[0][4][0][i_tie + 1]) == list(range(N - 1, N - 1 - K, -1))       # nothing known: the highest indexes
    for x in indexes:
        x.close()
    for m in models:
        m.close()


def gpu_corpus(torch, n_docs, seed, vocab=50_000):
    """SURVEY.md 8(d)'s BM25 corpus sampled on the GPU (numpy's Zipf sampler needs ~5 min for 1.5e9 tokens); the
    same generator as bench.py's."""
    import bench

    bench.BM25_VOCAB = vocab
    return bench.gen_bm25_corpus(np, torch, torch.device("cuda", 0), n_docs, seed)


def grouped_on_device(torch, indptr, toks, vocab):
    """bench.py's device-sorted stand-in for oracle.bm25.group_postings (integer bookkeeping only)."""
    import bench

    return bench.bm25_grouping_on_device(np, torch, torch.device("cuda", 0), indptr, toks, vocab)


def test_grouping_on_device_equals_the_oracles_own(corpus):
    import torch

    from oracle import bm25 as ob

    rows, indptr, toks, qvecs, qterms, cut = corpus
    for a, b in zip(grouped_on_device(torch, indptr, toks, VOCAB), ob.group_postings(indptr, toks, VOCAB)):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n_docs,n_queries", [(1_250_000, 64), (10_000_000, 16)])
def test_bm25_at_c4_sizes_bit_identical(n_docs, n_queries):
    """C4's BM25 leg at its per-GPU size (10M chunks / 8 GPUs) and at the whole corpus on one GPU (it fits): scores and
    top-10 of a query sample (SURVEY 8(d) mix: rare, mid-band, frequent, out-of-vocabulary, repeated terms) bit-identical
    to the CSR restatement of rank-bm25."""
    import torch

    import bench
    from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
    from oracle import bm25 as ob

    vocab = 50_000
    indptr, toks = gpu_corpus(torch, n_docs, 4242)
    dev = DeviceBM25.from_token_ids(indptr, toks, vocab)
    o = ob.BM25OkapiCSR(indptr, toks, vocab, grouped=grouped_on_device(torch, indptr, toks, vocab))
    info = dev.info()
    assert info["n_postings"] == len(o.t_doc) and info["avgdl"] == o.avgdl and info["average_idf"] == o.average_idf
    np.testing.assert_array_equal(dev.idf(), o.idf)
    qs = bench.bm25_queries(np, n_queries, 31)
    qs[2] = [0, 1, 2]  # the three most frequent terms: millions of postings
    idx, sc, cnt = dev.search(qs, 10)
    for i, q in enumerate(qs):
        s_all = o.get_scores(q)
        want = ob.top_n_indexes(s_all, 10)
        np.testing.assert_array_equal(idx[i], want, err_msg=f"query {i} {q}")
        np.testing.assert_array_equal(sc[i], s_all[want])
    np.testing.assert_array_equal(dev.get_scores(qs[0]), o.get_scores(qs[0]))
    dev.close()
