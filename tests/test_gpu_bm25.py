"""GPU parity of the BM25 path (C ABI -> HIP) against the oracle restatement of rank-bm25 0.2.2.
Scores are compared BIT-EXACT (float64, same operation order); top-n compared exactly, including the
reference's reversed stable tie-break."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bm25 as ob  # noqa: E402


@pytest.fixture(scope="module")
def br():
    from aidial_rag_amd import _native
    from aidial_rag_amd.retrievers import bm25_retriever

    assert _native.device_count() >= 1
    return bm25_retriever


CORPUS = [
    ["hello", "there", "good", "man"],
    ["it", "is", "quite", "windy", "in", "london"],
    ["how", "is", "the", "weather", "today"],
    [],
    ["is", "is", "is", "london"],
]


class Rec:
    def __init__(self, docs):
        self.text_index = None if docs is None else [type("I", (), {"chunk_index": i, "tokenized_text": t})() for i, t in enumerate(docs)]


def test_small_corpus_scores_and_topn(br):
    r = br.BM25Retriever.from_doc_records([Rec(CORPUS[:2]), Rec(None), Rec(CORPUS[2:])], k=3, preprocess=str.split)
    o = ob.build(CORPUS)
    info = r.bm25.info()
    assert info["n_docs"] == 5 and info["avgdl"] == o.avgdl and info["average_idf"] == o.average_idf
    idf = r.bm25.idf()
    assert len(idf) == len(o.idf)  # r.vocab is process-wide; the model's own term ids cover exactly this corpus
    for w, v in o.idf.items():
        assert idf[r.term_id(w)] == v
    for q in (["windy", "london", "nope", "london"], ["is"], ["absent"], [], ["london"], ["is", "is", "hello"]):
        got = r.bm25.get_scores(r._ids(q))
        np.testing.assert_array_equal(got, o.get_scores(q))
        for n in (1, 3, 5, 9):
            np.testing.assert_array_equal(r._get_top_n_indexes(q, n), ob.top_n_indexes(o.get_scores(q), n))
    docs = r._get_relevant_documents("absent words only")
    assert [(d.metadata["doc_id"], d.metadata["chunk_id"]) for d in docs] == [(2, 2), (2, 1), (2, 0)]
    assert r.get_relevant_documents_batch(["london is", "hello"]) == [r._get_relevant_documents("london is"), r._get_relevant_documents("hello")]


def test_empty_index_raises_like_reference(br):
    with pytest.raises(ValueError, match="Text index is empty."):
        br.BM25Retriever.from_doc_records([Rec([[], []]), Rec(None)], k=3)
    assert br.BM25Retriever.has_index([Rec([[], []])]) is False
    assert br.BM25Retriever.has_index([Rec(CORPUS)]) is True


def synth(n, vocab, seed, mean_len=150):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.round(rng.normal(mean_len, mean_len * 0.27, n)), 1, 400).astype(np.int64)
    lens[rng.random(n) < 0.001] = 0
    lens[::5003] = 0
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.minimum(rng.zipf(1.07, int(lens.sum())) - 1, vocab - 1).astype(np.int32)
    return indptr, toks


def queries(vocab, nq, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(nq):
        q = list(rng.integers(0, vocab, rng.integers(2, 9)))
        if i % 2:
            q = list(rng.integers(50, min(vocab, 5000), len(q)))  # mid-frequency band
        if i % 20 == 3:
            q[0] = vocab + 7  # out of vocabulary
        if i % 20 == 5:
            q.append(q[0])  # repeated term
        if i % 20 == 7:
            q = [0, 1]  # very frequent terms (negative-idf floor)
        out.append([int(t) for t in q])
    return out


@pytest.mark.parametrize("n,vocab", [(3000, 500), (20000, 5000), (100_000, 50_000)])
def test_synthetic_vs_oracle_bit_exact(br, n, vocab):
    indptr, toks = synth(n, vocab, 777 + n)
    dev = br.DeviceBM25.from_token_ids(indptr, toks, vocab)
    o = ob.BM25OkapiCSR(indptr, toks, vocab)
    assert dev.info()["avgdl"] == o.avgdl and dev.info()["average_idf"] == o.average_idf
    np.testing.assert_array_equal(dev.idf(), o.idf)
    qs = queries(vocab, 40, 778)
    idx, sc, cnt = dev.search(qs, 10)
    for i, q in enumerate(qs):
        want = o.get_scores(q)
        if i < 6:
            np.testing.assert_array_equal(dev.get_scores(q), want)
        top = ob.top_n_indexes(want, 10)
        assert cnt[i] == 10
        np.testing.assert_array_equal(idx[i], top, err_msg=f"query {i} {q}")
        np.testing.assert_array_equal(sc[i], want[top])
    # a query longer than the fast path's term table goes through the dense pass, batched with short ones
    long_q = [int(t) for t in np.random.default_rng(5).integers(0, vocab, 45)]
    i3, s3, c3 = dev.search([qs[1], long_q, qs[2]], 10)
    want = o.get_scores(long_q)
    np.testing.assert_array_equal(i3[1], ob.top_n_indexes(want, 10))
    np.testing.assert_array_equal(s3[1], want[i3[1]])
    np.testing.assert_array_equal(i3[0], idx[1])
    np.testing.assert_array_equal(i3[2], idx[2])
    # k larger than a tile's population and than the corpus
    small = br.DeviceBM25.from_token_ids(indptr[:8], toks[: indptr[7]], vocab)
    i2, s2, c2 = small.search([qs[0]], 64)
    assert c2[0] == 7
    # any n (bm25_retriever.py:81-84): beyond 64 the dense scores are ranked in rounds of 64; the zero tail and its
    # reversed tie-break included
    for kk in (65, 200, n + 10):
        ib, sb, cb = dev.search(qs[:5] + [[], [vocab + 3]], kk)
        for i, q in enumerate(qs[:5] + [[], [vocab + 3]]):
            want = o.get_scores(q)
            top = ob.top_n_indexes(want, kk)
            assert cb[i] == len(top) == min(kk, n)
            np.testing.assert_array_equal(ib[i, : cb[i]], top, err_msg=f"k={kk} query {i}")
            np.testing.assert_array_equal(sb[i, : cb[i]], want[top])


def _check_batch(dev, o, qs, k=10):
    idx, sc, cnt = dev.search(qs, k)
    for i, q in enumerate(qs):
        want = o.get_scores(q)
        top = ob.top_n_indexes(want, k)
        assert cnt[i] == k
        np.testing.assert_array_equal(idx[i], top, err_msg=f"query {i} {q}")
        np.testing.assert_array_equal(sc[i], want[top])


def test_topk_overflow_paths(br):
    """Orders that defeat the sampled bound of the block top-k: scores ascending with the document
    index (the 256-document sample is the worst of the tile, so nearly everything passes the bound),
    and a mass of exactly equal scores (every document is a finalist; ties go to the highest index).
    Both must fall back to the round-based selection and stay exact, also deep inside a batch."""
    n, vocab = 20000, 64
    docs = []
    for i in range(n):
        tf = 1 + (i % 8192) // 64
        docs.append([0] * tf + [1 + i % 40] * (1 + i % 3))
    lens = np.array([len(d) for d in docs], np.int64)
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.concatenate([np.asarray(d, np.int32) for d in docs])
    dev = br.DeviceBM25.from_token_ids(indptr, toks, vocab)
    o = ob.BM25OkapiCSR(indptr, toks, vocab)
    qs = [[0], [0, 5], [0, 0], [7, 0, 9]] * 24
    _check_batch(dev, o, qs)
    _check_batch(dev, o, qs[:3], k=64)
    # identical documents: all scores equal
    same = [[1, 2, 2]] * 9000 + [[3], [4, 4]] * 50
    lens = np.array([len(d) for d in same], np.int64)
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.concatenate([np.asarray(d, np.int32) for d in same])
    dev = br.DeviceBM25.from_token_ids(indptr, toks, 8)
    o = ob.BM25OkapiCSR(indptr, toks, 8)
    _check_batch(dev, o, [[1], [2, 1], [3, 1], [4]] * 10)


@pytest.mark.parametrize("batch", [1024])
def test_frequent_terms_large_batch(br, batch):
    """Very frequent terms (every tile lists more distinct documents than the candidate list holds)
    across a batch large enough that each workgroup walks many queries."""
    indptr, toks = synth(60000, 3000, 21)
    dev = br.DeviceBM25.from_token_ids(indptr, toks, 3000)
    o = ob.BM25OkapiCSR(indptr, toks, 3000)
    rng = np.random.default_rng(22)
    qs = [[int(t) for t in rng.integers(0, 40, rng.integers(1, 7))] for _ in range(batch)]
    idx, sc, cnt = dev.search(qs, 10)
    for i in range(0, batch, 37):
        want = o.get_scores(qs[i])
        top = ob.top_n_indexes(want, 10)
        np.testing.assert_array_equal(idx[i], top, err_msg=f"query {i} {qs[i]}")
        np.testing.assert_array_equal(sc[i], want[top])


@pytest.mark.parametrize("qc", [1, 2, 3, 64])
def test_short_pipelines_long_query_queues(br, qc):
    """The shape of the round-1 development fault (gpurun_out/f2.log: grid (123 tiles, ceil(4096 / 3)) = 3 queries per
    workgroup, 4096 queries of very frequent terms): the fast pass with its pipeline depth pinned
    to 1, 2, 3 (prologue and epilogue overlap; tails of 1 and 2 queries: 4099 = 3 * 1366 + 1) and to the maximum,
    over 8 tiles with every tile listing more touched documents than the candidate list holds, out-of-vocabulary
    and repeated terms mixed in.  Bit-identical to the oracle on a strided sample and across the pinned depths."""
    indptr, toks = synth(60000, 3000, 21)
    dev = br.DeviceBM25.from_token_ids(indptr, toks, 3000)
    o = ob.BM25OkapiCSR(indptr, toks, 3000)
    rng = np.random.default_rng(23)
    batch = 4099
    qs = [[int(t) for t in rng.integers(0, 40, rng.integers(1, 7))] for _ in range(batch)]
    for i in range(5, batch, 97):
        qs[i][0] = 3000 + i  # out of vocabulary
    for i in range(11, batch, 101):
        qs[i] = qs[i] + qs[i][:1]  # repeated term
    qs[batch - 1] = []  # an empty query at the very end of the queue
    auto = dev.search(qs, 10)
    dev.tune(qc)
    try:
        idx, sc, cnt = dev.search(qs, 10)
    finally:
        dev.tune(0)
    np.testing.assert_array_equal(idx, auto[0])
    np.testing.assert_array_equal(sc, auto[1])
    np.testing.assert_array_equal(cnt, auto[2])
    for i in list(range(0, batch, 41)) + [batch - 3, batch - 2, batch - 1]:
        want = o.get_scores(qs[i])
        top = ob.top_n_indexes(want, 10)
        np.testing.assert_array_equal(idx[i], top, err_msg=f"qc={qc} query {i} {qs[i]}")
        np.testing.assert_array_equal(sc[i], want[top])
    with pytest.raises(ValueError):
        dev.tune(65)


def test_device_buffer_api_matches_host_api(br):
    """mir_bm25_search_device: queries, results and scratch in HBM, asynchronous on the caller's stream."""
    import torch

    indptr, toks = synth(30000, 4000, 31)
    dev = br.DeviceBM25.from_token_ids(indptr, toks, 4000)
    qs = queries(4000, 50, 32)
    want_idx, want_sc, want_cnt = dev.search(qs, 10)
    ptr = np.zeros(len(qs) + 1, np.int32)
    ptr[1:] = np.cumsum([len(q) for q in qs])
    flat = np.concatenate([np.asarray(q, np.int32) for q in qs])
    d_terms = torch.from_numpy(flat).cuda()
    d_ptr = torch.from_numpy(ptr).cuda()
    o_idx = torch.zeros((len(qs), 10), dtype=torch.int64, device="cuda")
    o_sc = torch.zeros((len(qs), 10), dtype=torch.float64, device="cuda")
    o_cnt = torch.zeros(len(qs), dtype=torch.int32, device="cuda")
    ws = torch.empty(dev.workspace_bytes(len(qs), 10), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):  # the scratch is reusable
        dev.search_device(d_terms.data_ptr(), d_ptr.data_ptr(), len(qs), 10, o_idx.data_ptr(), o_sc.data_ptr(),
                          o_cnt.data_ptr(), ws.data_ptr(), stream=st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o_idx.cpu().numpy(), want_idx)
    np.testing.assert_array_equal(o_sc.cpu().numpy(), want_sc)
    np.testing.assert_array_equal(o_cnt.cpu().numpy(), want_cnt)


def test_dict_loop_oracle_agrees_on_token_ids(br):
    indptr, toks = synth(400, 200, 5, mean_len=30)
    corpus = [toks[indptr[i] : indptr[i + 1]].tolist() for i in range(400)]
    o = ob.BM25Okapi(corpus)
    dev = br.DeviceBM25.from_token_ids(indptr, toks, 200)
    for q in ([0], [1, 5, 9], [3, 3, 150], [199, 7, 250, -1]):
        np.testing.assert_array_equal(dev.get_scores(q), o.get_scores(q))


def test_sharded_stats_override(br):
    """Two document shards with GLOBAL idf / avgdl reproduce the unsharded model."""
    indptr, toks = synth(20000, 3000, 9)
    full = br.DeviceBM25.from_token_ids(indptr, toks, 3000)
    idf, avgdl = full.idf(), full.info()["avgdl"]
    cut = 9000
    a = br.DeviceBM25.from_token_ids(indptr[: cut + 1], toks[: indptr[cut]], 3000, idf=idf, avgdl=avgdl)
    b = br.DeviceBM25.from_token_ids(indptr[cut:] - indptr[cut], toks[indptr[cut] :], 3000, idf=idf, avgdl=avgdl, doc_offset=cut)
    qs = queries(3000, 12, 10)
    for q in qs:
        np.testing.assert_array_equal(np.concatenate([a.get_scores(q), b.get_scores(q)]), full.get_scores(q))
    ia, sa, ca = a.search(qs, 10)
    ib, sb, cb = b.search(qs, 10)
    fi, fs, fc = full.search(qs, 10)
    from aidial_rag_amd import _native as nat

    dist = np.ascontiguousarray(np.stack([sa, sb]))
    rows = np.ascontiguousarray(np.stack([ia, ib]))
    cnts = np.ascontiguousarray(np.stack([ca, cb]))
    od, orow, oc = np.zeros_like(sa), np.zeros_like(ia), np.zeros_like(ca)
    nat.check(nat.lib.mir_topk_merge_host(nat.ptr(dist), nat.ptr(rows), nat.ptr(cnts), 2, 0, len(qs), 10, 1, nat.ptr(od), nat.ptr(orow), nat.ptr(oc)))
    np.testing.assert_array_equal(orow, fi)
    np.testing.assert_array_equal(od, fs)


def test_sharded_build_then_global_stats(br):
    """The sharded build of config C4 (retrievers/sharded_bm25.py) on one GPU: three document shards built with
    placeholder statistics (one of them EMPTY of tokens), their corpus statistics combined as the all-reduce would,
    the library's idf routine, global statistics installed on the device (posting weights re-derived for the global
    avgdl) -> scores and idf bit-identical to the unsharded model; ShardedBM25 (world 1) over device buffers agrees."""
    import torch

    from aidial_rag_amd.retrievers import sharded_bm25 as sbm

    indptr, toks = synth(20000, 3000, 9)
    indptr = np.concatenate([indptr, np.full(50, indptr[-1])])  # 50 empty documents at the end: the third shard
    n = len(indptr) - 1
    full = br.DeviceBM25.from_token_ids(indptr, toks, 3000)
    cuts = [0, 9000, 20000, n]
    shards, stats = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sh = br.DeviceBM25.from_token_ids(indptr[lo : hi + 1] - indptr[lo], toks[indptr[lo] : indptr[hi]], 3000,
                                          idf=np.zeros(3000), avgdl=1.0, doc_offset=lo)
        shards.append(sh)
        stats.append(sh.corpus_stats())
    assert stats[2][2] == 0 and stats[2][3] == 50
    imax = np.iinfo(np.int64).max
    df = sum(s[0] for s in stats)
    offs = np.cumsum([0] + [s[2] for s in stats])
    first = np.min([np.where(s[1] == imax, imax, s[1] + o) for s, o in zip(stats, offs)], axis=0)
    total, n_docs = sum(s[2] for s in stats), sum(s[3] for s in stats)
    idf, avg_idf = sbm.idf_from_stats(df, first, n_docs)
    np.testing.assert_array_equal(idf, full.idf())
    assert avg_idf == full.info()["average_idf"] and total / n_docs == full.info()["avgdl"]
    for sh in shards:
        sh.set_global_stats(idf, total / n_docs, avg_idf)
        assert sh.info()["avgdl"] == full.info()["avgdl"]
    qs = queries(3000, 12, 10)
    for q in qs:
        np.testing.assert_array_equal(np.concatenate([sh.get_scores(q) for sh in shards]), full.get_scores(q))
    # ShardedBM25 with a single rank over the unsharded model: device-buffer search + device merge
    s1 = sbm.ShardedBM25(local_model=full)
    flat = torch.tensor(np.concatenate([np.asarray(q, np.int32) for q in qs]), dtype=torch.int32, device="cuda")
    ptr = torch.tensor(np.concatenate(([0], np.cumsum([len(q) for q in qs]))), dtype=torch.int32, device="cuda")
    sc, idx, cnt = s1.search(flat, 10, ptr)
    torch.cuda.synchronize()
    fi, fs, fc = full.search(qs, 10)
    np.testing.assert_array_equal(idx.cpu().numpy(), fi)
    np.testing.assert_array_equal(sc.cpu().numpy(), fs)


def test_full_size_1m_documents_50k_vocabulary(br):
    """BASELINE config C3 at full size (1M chunks, 50k-term vocabulary, SURVEY 8(d) corpus and query mix):
    float64 scores and top-10 bit-identical to the CSR restatement for a 64-query batch."""
    n, vocab = 1_000_000, 50_000
    rng = np.random.default_rng(777)
    lens = np.clip(np.round(rng.normal(150, 40, n)), 1, 400).astype(np.int64)
    lens[rng.random(n) < 0.001] = 0
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.minimum(rng.zipf(1.07, int(lens.sum())) - 1, vocab - 1).astype(np.int32)
    dev = br.DeviceBM25.from_token_ids(indptr, toks, vocab)
    o = ob.BM25OkapiCSR(indptr, toks, vocab)
    assert dev.info()["avgdl"] == o.avgdl and dev.info()["average_idf"] == o.average_idf
    qs = queries(vocab, 64, 778)
    idx, sc, cnt = dev.search(qs, 10)
    for i, q in enumerate(qs):
        want = o.get_scores(q)
        top = ob.top_n_indexes(want, 10)
        np.testing.assert_array_equal(idx[i], top, err_msg=f"query {i} {q}")
        np.testing.assert_array_equal(sc[i], want[top])


def test_concurrent_retriever_calls_share_passes(br):
    """BM25Retriever._get_relevant_documents from many threads: same answers as sequentially, fewer passes than calls."""
    import threading

    rng = np.random.default_rng(41)
    words = [f"w{i}" for i in range(300)]
    docs = [[words[j] for j in rng.integers(0, 300, rng.integers(5, 40))] for _ in range(20000)]
    r = br.BM25Retriever.from_doc_records([Rec(docs[:12000]), Rec(docs[12000:])], k=5, preprocess=str.split)
    qs = [" ".join(words[j] for j in rng.integers(0, 300, rng.integers(1, 5))) for _ in range(96)]
    key = lambda found: [(d.metadata["doc_id"], d.metadata["chunk_id"]) for d in found]
    want = [key(r._get_relevant_documents(q)) for q in qs]
    gc = r._commits[5]
    passes0 = gc.passes
    got = [None] * 96

    def work(t):
        for i in range(t, 96, 16):
            got[i] = key(r._get_relevant_documents(qs[i]))

    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == want
    assert gc.passes - passes0 < 96


def test_new_document_combination_reuses_token_ids(br):
    """A model over a NEW combination of documents whose token lists were seen before does no per-token Python:
    every document's term-id arrays are cached (kind "bm25doc") in a process-wide vocabulary.  Term ids the
    combination does not use must not disturb idf / average_idf: scores stay bit-identical to rank-bm25's."""
    from aidial_rag_amd.retrievers import _device_cache as dc

    rng = np.random.default_rng(5)
    words = [f"t{i}" for i in range(400)]

    def doc(n, lo, hi):
        return [[words[j] for j in rng.integers(lo, hi, rng.integers(0, 30))] for _ in range(n)]

    a, b, c = Rec(doc(60, 0, 200)), Rec(doc(5, 100, 400)), Rec(doc(90, 150, 300))
    dc.CACHE.clear()
    base = list(dc.CACHE.by_kind.get("bm25doc", [0, 0]))
    delta = lambda: [x - y for x, y in zip(dc.CACHE.by_kind["bm25doc"], base)]

    def check(recs):
        r = br.BM25Retriever.from_doc_records(recs, k=6, preprocess=str.split)
        corpus = [item.tokenized_text for rec in recs if rec.text_index is not None for item in rec.text_index]
        o = ob.build(corpus)
        info = r.bm25.info()
        assert info["n_docs"] == len(corpus) and info["avgdl"] == o.avgdl and info["average_idf"] == o.average_idf
        for q in (["t150", "t160", "t1"], ["t399"], ["t250", "t250", "t120"], ["never"], ["t5", "t180"]):
            np.testing.assert_array_equal(r.bm25.get_scores(r._ids(q)), o.get_scores(q))
            np.testing.assert_array_equal(r._get_top_n_indexes(q, 6), ob.top_n_indexes(o.get_scores(q), 6))
        return r

    check([a, b])
    assert delta() == [0, 2]
    r2 = check([c, Rec(None), a])  # a's ids come from the cache; c brings words a and b never had
    assert delta() == [1, 3]
    check([b, c])                  # nothing new at all
    assert delta() == [3, 3]
    docs = r2._get_relevant_documents("t299")
    assert docs and all(d.metadata["doc_id"] in (0, 2) for d in docs)
