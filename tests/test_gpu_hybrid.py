"""End to end on the GPU, BASELINE config 1 shape: ~1k synthetic chunks in several documents -> encoder ->
semantic index + BM25 + page (description) index -> weighted RRF, against the same pipeline assembled from the
oracle pieces on the SAME embedding matrix (encoder parity is measured separately; retrieval parity must not
depend on float16-vs-float32 encoder noise)."""

import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Chunk:
    def __init__(self, text, page):
        self.text, self.metadata = text, {"page_number": page}


class Record:
    def __init__(self, chunks, text_index, embeddings_index, description_embeddings_index=None, multimodal_embeddings_index=None):
        self.chunks, self.text_index, self.embeddings_index = chunks, text_index, embeddings_index
        self.description_embeddings_index, self.multimodal_embeddings_index = description_embeddings_index, multimodal_embeddings_index


class WordTokenizer:
    """Whitespace 'tokenizer' over a closed synthetic vocabulary (stands in for WordPiece)."""

    def __init__(self, words):
        self.ids = {w: 1000 + i for i, w in enumerate(words)}

    def __call__(self, texts, add_special_tokens=True, truncation=True, max_length=512):
        out = []
        for t in texts:
            ids = [101] + [self.ids.get(w, 100) for w in t.split()][: max_length - 2] + [102]
            out.append(ids)
        return {"input_ids": out}


@pytest.fixture(scope="module")
def world():
    from aidial_rag_amd import retrieval_chain
    from aidial_rag_amd.embeddings.embeddings import BgeEncoder
    from aidial_rag_amd.retrievers import bm25_retriever as br
    from aidial_rag_amd.retrievers import embeddings_index as ei
    from aidial_rag_amd.retrievers.semantic_retriever import SemanticRetriever
    from oracle import encoder as oe

    rng = np.random.default_rng(2024)
    words = [f"w{i}" for i in range(3000)]
    tok = WordTokenizer(words + "Represent this question for searching relevant passages:".split())
    enc = BgeEncoder.from_state_dict(oe.make_model(layers=12, seed=1, scale=2.0).state_dict(), tokenizer=tok)
    records = []
    zipf = np.minimum(rng.zipf(1.2, 200_000) - 1, len(words) - 1)
    pos = 0
    for n_chunks in (400, 1, 350, 250):
        chunks = []
        for c in range(n_chunks):
            L = int(rng.integers(20, 120))
            chunks.append(Chunk(" ".join(words[i] for i in zipf[pos : pos + L]), page=1 + c // 5))
            pos += L
        text_index = asyncio.run(br.BM25Retriever.build_index(chunks, preprocess=str.split))
        emb_index = asyncio.run(SemanticRetriever.build_index(chunks, encoder=enc))
        n_pages = 1 + (n_chunks - 1) // 5
        page_vecs = enc.embed_documents_numpy([" ".join(words[i] for i in rng.integers(0, 3000, 30)) for _ in range(n_pages)])
        desc = ei.pack_multi_embeddings(list(range(n_pages)), page_vecs, n_pages) if n_chunks > 1 else None
        records.append(Record(chunks, text_index, emb_index, description_embeddings_index=desc))
    return retrieval_chain, enc, records, tok, rng, words


def oracle_pipeline(records, enc, query, words):
    from aidial_rag_amd.embeddings.embeddings import BGE_QUERY_INSTRUCTION_EN
    from oracle import bm25 as ob
    from oracle import embeddings_index as oi
    from oracle import fusion as of

    q = np.array(enc.embed_query(query))  # same query vector (float64 list -> array), as semantic_retriever.py:49
    sem, _ = oi.find(q, [oi.create_index_by_chunk([it.embeddings for it in r.embeddings_index]) for r in records], "sqeuclidean_dist", 7)
    corpus, owners = [], []
    for i, r in enumerate(records):
        for it in r.text_index:
            corpus.append(it.tokenized_text)
            owners.append((i, it.chunk_index))
    top = ob.top_n_indexes(ob.build(corpus).get_scores(query.split()), 7)
    bm = [owners[int(t)] for t in top]
    desc_idx = [
        oi.create_index_by_page([c.metadata["page_number"] for c in r.chunks],
                                None if r.description_embeddings_index is None else [it.embeddings for it in r.description_embeddings_index])
        for r in records
    ]
    desc, _ = oi.find(q, desc_idx, "sqeuclidean_dist", 7)
    return sem, bm, desc, of.weighted_reciprocal_rank([sem, bm, desc], [1.0, 1.0, 1.0])


def test_hybrid_retrieval_equals_oracle_pipeline(world):
    retrieval_chain, enc, records, tok, rng, words = world
    assert sum(len(r.chunks) for r in records) == 1001
    retr = retrieval_chain.create_retriever(records, encoder=enc, keywords_preprocess=str.split)
    assert len(retr.retrievers) == 3
    for qi in range(12):
        query = " ".join(words[i] for i in rng.integers(0, 300 if qi % 2 else 3000, rng.integers(3, 9)))
        sem, bm, desc, fused = oracle_pipeline(records, enc, query, words)
        key = lambda docs: [(d.metadata["doc_id"], d.metadata["chunk_id"]) for d in docs]  # noqa: E731
        assert key(retr.retrievers[0].invoke(query)) == sem
        assert key(retr.retrievers[1].invoke(query)) == bm
        assert key(retr.retrievers[2].invoke(query)) == desc
        got = retr.invoke(query)
        assert key(got) == fused
        assert {d.metadata["retrieval_type"].value for d in retr.retrievers[2].invoke(query)} == {"image"}


def test_semantic_batch_equals_single(world):
    retrieval_chain, enc, records, tok, rng, words = world
    from aidial_rag_amd.retrievers.semantic_retriever import SemanticRetriever

    r = SemanticRetriever.from_doc_records(records, 7, encoder=enc)
    qs = [" ".join(words[i] for i in rng.integers(0, 3000, 6)) for _ in range(5)]
    assert r.get_relevant_documents_batch(qs) == [r.invoke(q) for q in qs]
    docs = asyncio.run(r._aget_relevant_documents(qs[0]))
    assert docs == r.invoke(qs[0])


def test_indexes_survive_across_requests():
    """a second `from_doc_records` over the SAME record objects reuses the indexes already
    in HBM (vector index and BM25 model); different records build new ones; results are unaffected."""
    import numpy as np

    from aidial_rag_amd.retrievers import _device_cache as dc
    from aidial_rag_amd.retrievers.bm25_retriever import BM25Retriever
    from aidial_rag_amd.retrievers.embeddings_index import ItemEmbeddings
    from aidial_rag_amd.retrievers.semantic_retriever import SemanticRetriever

    rng = np.random.default_rng(3)

    class Item:
        def __init__(self, i, toks):
            self.chunk_index, self.tokenized_text = i, toks

    class Rec:
        def __init__(self, n):
            v = rng.standard_normal((n, 384)).astype(np.float32)
            self.embeddings_index = [ItemEmbeddings(v[i : i + 1]) for i in range(n)]
            self.text_index = [Item(i, [f"w{j}" for j in rng.integers(0, 50, 8)]) for i in range(n)]

    recs = [Rec(300), Rec(200)]
    dc.CACHE.clear()
    kinds = ("vector", "rows", "bm25", "bm25doc")
    base = {k: list(dc.CACHE.by_kind.get(k, [0, 0])) for k in kinds}
    stat = lambda: {k: [x - y for x, y in zip(dc.CACHE.by_kind.get(k, [0, 0]), base[k])] for k in kinds}  # [hits, misses]
    q = rng.standard_normal(384)
    s1 = SemanticRetriever.from_doc_records(recs, k=5)
    r1 = s1._find_relevant_documents(q)
    b1 = BM25Retriever.from_doc_records(recs, k=4, preprocess=str.split)
    k1 = b1._get_relevant_documents("w1 w7 w9")
    # first sight: the vector index and its two row blocks, the BM25 model and the two documents' token-id arrays
    assert stat() == {"vector": [0, 1], "rows": [0, 2], "bm25": [0, 1], "bm25doc": [0, 2]}
    s2 = SemanticRetriever.from_doc_records(recs, k=5)
    b2 = BM25Retriever.from_doc_records(recs, k=4, preprocess=str.split)
    assert s2._find_relevant_documents(q) == r1 and b2._get_relevant_documents("w1 w7 w9") == k1
    assert s2.index._device_index() is s1.index._device_index() and b2.bm25 is b1.bm25
    assert stat() == {"vector": [1, 1], "rows": [0, 2], "bm25": [1, 1], "bm25doc": [0, 2]}  # nothing below the top level is touched
    other = [Rec(100)]
    s3 = SemanticRetriever.from_doc_records(other, k=5)
    assert s3.index._device_index() is not s1.index._device_index()
    assert stat()["vector"] == [1, 2] and stat()["rows"] == [0, 3]
    # a tiny budget keeps only the most recent entry; evicted indexes stay valid for whoever still holds them
    small = dc.DeviceCache(budget_bytes=1)
    a = small.get_or_build("x", 0, [recs[0]], lambda: ("A", 10))
    b = small.get_or_build("x", 0, [recs[1]], lambda: ("B", 10))
    assert (a, b) == ("A", "B") and len(small) == 1
    assert s1._find_relevant_documents(q) == r1


def test_new_document_combinations_are_composed_in_hbm():
    """A request over a NEW combination of documents uploads only the documents not seen before: every document's
    rows are a block in HBM (mir_rows) and the index is concatenated device-to-device (mir_index_create_from_rows).
    The composed index answers exactly like one uploaded flat from the host - ids, order, distances, ties."""
    import numpy as np

    from aidial_rag_amd.retrievers import _device_cache as dc
    from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex, DeviceRows, DocIndex, EmbeddingsIndex, ItemEmbeddings
    from aidial_rag_amd.retrievers.semantic_retriever import SemanticRetriever

    rng = np.random.default_rng(11)

    class Rec:
        def __init__(self, n):
            v = rng.standard_normal((n, 384)).astype(np.float32)
            if n > 4:
                v[3] = v[1]  # duplicate rows: ties inside a document
            self.embeddings_index = [ItemEmbeddings(v[i : i + 1]) for i in range(n)]
            self.text_index = None

    a, b, c, empty = Rec(700), Rec(33), Rec(1500), Rec(0)
    c.embeddings_index[7] = ItemEmbeddings(np.asarray(a.embeddings_index[1].embeddings))  # ... and across documents
    dc.CACHE.clear()
    rows0 = list(dc.CACHE.by_kind.get("rows", [0, 0]))
    qs = rng.standard_normal((9, 384))
    first = SemanticRetriever.from_doc_records([a, b], k=7)
    r_first = [first._find_relevant_documents(q) for q in qs]
    assert [x - y for x, y in zip(dc.CACHE.by_kind["rows"], rows0)] == [0, 2]
    second = SemanticRetriever.from_doc_records([c, empty, a], k=7)  # a is known, c is new, the empty one has no block
    r_second = [second._find_relevant_documents(q) for q in qs]
    assert [x - y for x, y in zip(dc.CACHE.by_kind["rows"], rows0)] == [1, 3]

    def flat(recs):  # no cache_sources: flattened on the host, uploaded as one matrix
        from aidial_rag_amd.retrievers.embeddings_index import create_index_by_chunk
        from aidial_rag_amd.index_record import RetrievalType

        ix = EmbeddingsIndex(RetrievalType.TEXT, [create_index_by_chunk(r.embeddings_index) for r in recs], limit=7)
        return [ix.find(q) for q in qs]

    assert r_first == flat([a, b]) and r_second == flat([c, a])  # from_doc_records drops records without an index
    # an empty DocIndex in the middle keeps its position (embeddings_index.py:67-69): doc ids 0 and 2, no block for it
    from aidial_rag_amd.index_record import RetrievalType
    from aidial_rag_amd.retrievers.embeddings_index import create_index_by_chunk

    marker = object()
    holes = EmbeddingsIndex(RetrievalType.TEXT, [create_index_by_chunk(c.embeddings_index), DocIndex(),
                                                 create_index_by_chunk(a.embeddings_index)], limit=7,
                            cache_sources=[c.embeddings_index, marker, a.embeddings_index])
    r_holes = [holes.find(q) for q in qs]
    assert r_holes == flat([c, empty, a])
    assert {d.metadata["doc_id"] for r in r_holes for d in r} == {0, 2}
    assert [x - y for x, y in zip(dc.CACHE.by_kind["rows"], rows0)] == [3, 3]  # both blocks were already in HBM

    # C ABI level, float16 blocks of unequal sizes, every metric: identical arrays to the flat upload
    parts_host = [rng.standard_normal((n, 840)).astype(np.float16) for n in (5, 1000, 1, 64, 2049)]
    chunk = [rng.integers(0, 1 << 40, len(p)) for p in parts_host]
    blocks = [DeviceRows.from_host(p, ci) for p, ci in zip(parts_host, chunk)]
    doc_ids = [4, 0, 9, 9, 2]
    composed = DeviceIndex.from_rows(blocks, doc_ids)
    whole = DeviceIndex.from_host(np.concatenate(parts_host), np.concatenate(chunk),
                                  np.concatenate([np.full(len(p), d, np.int32) for p, d in zip(parts_host, doc_ids)]))
    q16 = rng.standard_normal((5, 840))
    for metric in ("sqeuclidean_dist", "cosine_sim", "inner_product", "euclidean_dist"):
        for got, want in zip(composed.search(q16, 10, metric), whole.search(q16, 10, metric)):
            np.testing.assert_array_equal(got, want)
    for blk in blocks:
        blk.close()  # blocks are copied at composition: the index outlives them
    for got, want in zip(composed.search(q16, 10, "cosine_sim"), whole.search(q16, 10, "cosine_sim")):
        np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        DeviceIndex.from_rows([DeviceRows.from_host(np.zeros((2, 8), np.float32)), DeviceRows.from_host(np.zeros((2, 9), np.float32))])


def test_row_block_edges():
    """mir_rows / mir_index_create_from_rows edge cases: an empty block in the middle, chunk ids defaulting to
    0..n-1, float16 blocks that are widened (d <= 512), a single one-row block, and argument errors."""
    import numpy as np

    from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex, DeviceRows

    rng = np.random.default_rng(21)
    a = rng.standard_normal((70, 48)).astype(np.float32)
    c = rng.standard_normal((1, 48)).astype(np.float32)
    empty = np.zeros((0, 48), np.float32)
    blocks = [DeviceRows.from_host(a), DeviceRows.from_host(empty), DeviceRows.from_host(c)]
    assert [b.n for b in blocks] == [70, 0, 1] and blocks[1].hbm_bytes() == 0
    ix = DeviceIndex.from_rows(blocks)  # doc ids default to the block's position, chunk ids to the row within it
    q = rng.standard_normal((3, 48))
    doc, chunk, row, dist, cnt, _ = ix.search(q, 5, "sqeuclidean_dist")
    whole = DeviceIndex.from_host(np.concatenate([a, c]), np.concatenate([np.arange(70), np.arange(1)]),
                                  np.concatenate([np.zeros(70, np.int32), np.full(1, 2, np.int32)]))
    for got, want in zip((doc, chunk, row, dist, cnt), whole.search(q, 5, "sqeuclidean_dist")):
        np.testing.assert_array_equal(got, want)
    # one block of one row
    one = DeviceIndex.from_rows([blocks[2]], [7])
    d1, c1, _, _, n1, _ = one.search(q[:1], 3, "cosine_sim")
    assert n1[0] == 1 and d1[0, 0] == 7 and c1[0, 0] == 0
    # float16 blocks below the float16-native width: widened exactly, same answers as the flat float16 upload
    h = [rng.standard_normal((n, 96)).astype(np.float16) for n in (33, 5)]
    comp = DeviceIndex.from_rows([DeviceRows.from_host(x) for x in h])
    flat = DeviceIndex.from_host(np.concatenate(h), np.concatenate([np.arange(33), np.arange(5)]),
                                 np.concatenate([np.zeros(33, np.int32), np.ones(5, np.int32)]))
    q96 = rng.standard_normal((4, 96))
    for got, want in zip(comp.search(q96, 6, "inner_product"), flat.search(q96, 6, "inner_product")):
        np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        DeviceIndex.from_rows([])
    with pytest.raises(ValueError):
        DeviceRows.from_host(np.zeros((3, 4), np.float32), chunk_ids=[1, 2])
    with pytest.raises(ValueError):
        DeviceIndex.from_rows([blocks[0], DeviceRows.from_host(np.zeros((2, 48), np.float16))])  # dtype mismatch


@pytest.mark.parametrize("metric", ["sqeuclidean_dist", "cosine_sim"])
def test_multimodal_retriever_equals_oracle(metric):
    """MultimodalRetriever.from_doc_records (multimodal_retriever.py:96-153 upstream): page embeddings of a remote
    multimodal model (float32, d = 1024, NOT normalised), expanded page -> chunks by create_index_by_page, metric
    from the index config (sqeuclidean_dist default or cosine_sim, :55-63).  Against oracle.find over
    oracle.create_index_by_page on the same arrays: identical (doc_id, chunk_id) lists, k = 7 as the product
    (retrieval_chain.py:224) and k = 20."""
    from aidial_rag_amd.retrievers import embeddings_index as ei
    from aidial_rag_amd.retrievers.page_retrievers import MultimodalRetriever
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(31)
    d = 1024
    records, oracle_docs = [], []
    for n_pages, chunks_per_page in ((120, 9), (1, 3), (0, 0), (300, 5)):
        if n_pages == 0:
            records.append(Record([Chunk("x", 1)], None, None, multimodal_embeddings_index=None))
            oracle_docs.append(oi.DocIndex())
            continue
        pages = (rng.standard_normal((n_pages, d)) * rng.uniform(0.5, 3.0, (n_pages, 1))).astype(np.float32)
        chunks = [Chunk(f"c{c}", 1 + c // chunks_per_page) for c in range(n_pages * chunks_per_page)]
        mm = ei.pack_multi_embeddings(list(range(n_pages)), list(pages), n_pages)
        records.append(Record(chunks, None, None, multimodal_embeddings_index=mm))
        oracle_docs.append(oi.create_index_by_page([c.metadata["page_number"] for c in chunks], [it.embeddings for it in mm]))
    records[0].multimodal_embeddings_index[7].embeddings[0] = records[3].multimodal_embeddings_index[11].embeddings[0]  # cross-document tie
    oracle_docs[0] = oi.create_index_by_page([c.metadata["page_number"] for c in records[0].chunks],
                                             [it.embeddings for it in records[0].multimodal_embeddings_index])
    assert MultimodalRetriever.has_index(records)
    qvecs = {f"q{i}": rng.standard_normal(d).astype(np.float32).tolist() for i in range(6)}
    qvecs["tie"] = records[3].multimodal_embeddings_index[11].embeddings[0].tolist()
    for k in (7, 20):
        retr = MultimodalRetriever.from_doc_records(records, k=k, metric=metric, embed_query=qvecs.__getitem__)
        for name, vec in qvecs.items():
            got = retr.invoke(name)
            want, _ = oi.find(np.array(vec), oracle_docs, metric, k)
            assert [(x.metadata["doc_id"], x.metadata["chunk_id"]) for x in got] == want, (metric, k, name)
            assert all(x.metadata["retrieval_type"].value == "image" for x in got)
    with pytest.raises(RuntimeError):
        MultimodalRetriever.from_doc_records(records, k=1).invoke("no embedder")
