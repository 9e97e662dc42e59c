"""BM25 keyword retriever, same surface as aidial_rag/retrievers/bm25_retriever.py:26-114.

``BM25Retriever.from_doc_records(doc_records, k)`` flattens every document's
``text_index`` (items with ``chunk_index`` and ``tokenized_text``) in document
order, raises ``ValueError("Text index is empty.")`` when there is no token,
and builds the model; ``_get_relevant_documents(query)`` returns the top-k
``Document`` list.  The model lives on the GPU (``mir_bm25_create``): the host
keeps only the str -> term-id vocabulary; scoring and top-k are HIP kernels
with float64 arithmetic bit-identical to rank-bm25's.

Added over the reference: ``get_relevant_documents_batch`` (B queries per
launch); a single query is the B = 1 case of the same kernels.
"""

import asyncio
from concurrent.futures import ThreadPoolExecutor
import ctypes as C
import threading
from typing import Callable, Dict, Hashable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native as nat
from ..index_record import Document, RetrievalType, to_metadata_doc
from ._group_commit import _GroupCommit


class DeviceBM25:
    """Owner of one ``mir_bm25`` handle over token-id documents."""

    def __init__(self, handle, n_docs: int, vocab: int, device: int):
        self._h, self.n_docs, self.vocab, self.device = handle, n_docs, vocab, device

    @classmethod
    def from_token_ids(cls, indptr: np.ndarray, term_ids: np.ndarray, vocab: int, k1: float = 1.5, b: float = 0.75,
                       epsilon: float = 0.25, idf: Optional[np.ndarray] = None, avgdl: float = 0.0, device: int = 0,
                       doc_offset: int = 0):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        term_ids = np.ascontiguousarray(term_ids, dtype=np.int32)
        idf_c = None if idf is None else np.ascontiguousarray(idf, dtype=np.float64)
        if idf_c is not None and len(idf_c) != vocab:
            raise ValueError("idf override must have `vocab` entries")
        h = C.c_void_p()
        nat.check(nat.lib.mir_bm25_create(nat.ptr(indptr), nat.ptr(term_ids) if len(term_ids) else None, len(indptr) - 1,
                                          vocab, k1, b, epsilon, nat.ptr(idf_c), avgdl, device, doc_offset, C.byref(h)))
        return cls(h, len(indptr) - 1, vocab, device)

    def info(self) -> dict:
        n, v, p = C.c_int64(), C.c_int32(), C.c_int64()
        a, ai, hb = C.c_double(), C.c_double(), C.c_int64()
        nat.check(nat.lib.mir_bm25_info(self._h, C.byref(n), C.byref(v), C.byref(p), C.byref(a), C.byref(ai), C.byref(hb)))
        return {"n_docs": n.value, "vocab": v.value, "n_postings": p.value, "avgdl": a.value, "average_idf": ai.value,
                "hbm_bytes": hb.value}

    def corpus_stats(self):
        """-> (df int64[vocab], first_pos int64[vocab] (INT64_MAX = absent), total_tokens, n_docs) of THIS model's
        documents: what a document-sharded corpus all-reduces (retrievers/sharded_bm25.py)."""
        df = np.zeros(self.vocab, np.int64)
        first = np.zeros(self.vocab, np.int64)
        tot, n = C.c_int64(0), C.c_int64(0)
        nat.check(nat.lib.mir_bm25_corpus_stats(self._h, nat.ptr(df), nat.ptr(first), C.byref(tot), C.byref(n)))
        return df, first, int(tot.value), int(n.value)

    def set_global_stats(self, idf: np.ndarray, avgdl: float, average_idf: float = 0.0) -> None:
        """Install global idf[vocab] / avgdl (posting weights are re-derived on the device)."""
        idf = np.ascontiguousarray(idf, dtype=np.float64)
        if len(idf) != self.vocab:
            raise ValueError("idf must have `vocab` entries")
        nat.check(nat.lib.mir_bm25_set_global_stats(self._h, nat.ptr(idf), float(avgdl), float(average_idf)))

    def tune(self, queries_per_workgroup: int = 0) -> None:
        """Pin the fast pass's pipeline depth (1..64 queries per workgroup; 0 = per call).  Results do not depend on it."""
        nat.check(nat.lib.mir_bm25_tune(self._h, queries_per_workgroup))

    def idf(self) -> np.ndarray:
        out = np.zeros(self.vocab, np.float64)
        nat.check(nat.lib.mir_bm25_idf(self._h, nat.ptr(out)))
        return out

    def get_scores(self, query_ids: Sequence[int]) -> np.ndarray:
        q = np.ascontiguousarray(query_ids, dtype=np.int32)
        out = np.zeros(self.n_docs, np.float64)
        nat.check(nat.lib.mir_bm25_scores(self._h, nat.ptr(q) if len(q) else None, len(q), nat.ptr(out)))
        return out

    def search(self, queries_ids: Sequence[Sequence[int]], k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """-> (idx[b,k] i64, score[b,k] f64, count[b] i32), best first."""
        b = len(queries_ids)
        ptr = np.zeros(b + 1, np.int32)
        ptr[1:] = np.cumsum([len(q) for q in queries_ids])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(q, np.int32) for q in queries_ids]) if ptr[-1] else np.zeros(0, np.int32))
        idx = np.zeros((b, k), np.int64)
        sc = np.zeros((b, k), np.float64)
        cnt = np.zeros(b, np.int32)
        nat.check(nat.lib.mir_bm25_search(self._h, nat.ptr(flat) if len(flat) else None, nat.ptr(ptr), b, k, nat.ptr(idx),
                                          nat.ptr(sc), nat.ptr(cnt)))
        return idx, sc, cnt

    def workspace_bytes(self, b: int, k: int) -> int:
        """HBM scratch `search_device` needs for a batch of b queries."""
        n = int(nat.lib.mir_bm25_workspace_bytes(self._h, b, k))
        if n < 0:
            raise ValueError("bad batch shape")
        return n

    def search_device(self, q_terms_ptr: int, q_ptr_ptr: int, b: int, k: int, out_idx_ptr: int, out_score_ptr: int,
                      out_count_ptr: int, workspace_ptr: int, stream: int = 0) -> None:
        """`search` with every buffer in HBM (int32 term ids + int32 q_ptr[b+1] in, int64 idx[b,k] / float64
        score[b,k] / int32 count[b] out), asynchronous on `stream`."""
        nat.check(nat.lib.mir_bm25_search_device(self._h, q_terms_ptr or None, q_ptr_ptr, b, k, out_idx_ptr, out_score_ptr,
                                                 out_count_ptr, workspace_ptr, stream or None))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            nat.lib.mir_bm25_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TextIndexItem:
    """index_record.py:9-15: one chunk's token list."""

    __slots__ = ("chunk_index", "tokenized_text")

    def __init__(self, chunk_index: int, tokenized_text: List[str]):
        self.chunk_index = chunk_index
        self.tokenized_text = tokenized_text


# chunks per native call: large enough to occupy every host core (32 texts per work item), small enough that the Python
# side's token lists of one call stay cache-sized
_PREPROCESS_BATCH = 4096


def _build_text_index_chunks(chunks, preprocess: Optional[Callable[[str], List[str]]]) -> List[TextIndexItem]:
    # bm25_retriever.py:30-39.  Upstream maps keywords_preprocess over the chunks on a CPU pool, pure Python per token;
    # here the default preprocess is ONE native call per _PREPROCESS_BATCH chunks on all host cores with the GIL released
    # (keywords_search.keywords_preprocess_batch -> mir_keywords_preprocess), a caller's own callable is mapped as upstream.
    if preprocess is not None:
        return [TextIndexItem(chunk_index=i, tokenized_text=preprocess(chunk.text)) for i, chunk in enumerate(chunks)]
    from ..keywords_search import keywords_preprocess_batch

    out: List[TextIndexItem] = []
    starts = range(0, len(chunks), _PREPROCESS_BATCH)
    one = lambda b0: keywords_preprocess_batch([c.text for c in chunks[b0 : b0 + _PREPROCESS_BATCH]], lazy=True)
    if len(starts) <= 1:
        batches = map(one, starts)
    else:
        # two batches in flight: the native call of one (GIL released, all cores) runs beside the Python side of the other
        # (the TokenList views and the TextIndexItem objects: single-threaded by nature)
        pool = ThreadPoolExecutor(2, thread_name_prefix="mir-kwp")
        batches = pool.map(one, starts)
    for b0, toks in zip(starts, batches):
        out.extend(TextIndexItem(chunk_index=b0 + i, tokenized_text=t) for i, t in enumerate(toks))
    if len(starts) > 1:
        pool.shutdown(wait=False)
    return out


# str -> term id for every token this process has indexed (ids are dense in first-seen order; never reused)
_VOCAB: Dict[Hashable, int] = {}
_VOCAB_LOCK = threading.Lock()


def _doc_token_ids(text_index) -> Tuple[Tuple[np.ndarray, np.ndarray, np.ndarray], int]:
    """One document's `text_index` -> ((chunk_index i64[c], token count i64[c], term ids i32[tokens]), bytes held)."""
    chunk = np.fromiter((item.chunk_index for item in text_index), np.int64, len(text_index))
    lens = np.fromiter((len(item.tokenized_text) for item in text_index), np.int64, len(text_index))
    ids = np.empty(int(lens.sum()), np.int32)
    pos = 0
    from ..keywords_search import TokenList

    run = None  # (batch, first token, end token) of consecutive chunks of one native batch: ONE gather for the run

    def flush():
        nonlocal pos, run
        if run is not None:
            bt, a, b = run
            # the native preprocessor's views: the batch's DISTINCT tokens go through the vocabulary once, the chunks' ids are a gather
            np.take(bt.global_ids(_VOCAB, _VOCAB_LOCK), bt.ids[a:b], out=ids[pos : pos + (b - a)], mode="clip")  # (int32 table, int32 indexes, int32 out: no casts)
            pos += b - a
            run = None

    for item in text_index:
        toks = item.tokenized_text
        if isinstance(toks, TokenList):
            if run is not None and run[0] is toks.batch and run[2] == toks.a:
                run = (run[0], run[1], toks.b)
            else:
                flush()
                run = (toks.batch, toks.a, toks.b)
            continue
        flush()
        with _VOCAB_LOCK:
            look = _VOCAB.setdefault
            for t in toks:
                ids[pos] = look(t, len(_VOCAB))
                pos += 1
    flush()
    return (chunk, lens, ids), chunk.nbytes + lens.nbytes + ids.nbytes


class _FlatIndexTable:
    """flat chunk number -> (doc_index, chunk_index), the reference's `text_indexes` list held as two arrays."""

    def __init__(self, doc_of: np.ndarray, chunk_of: np.ndarray):
        self.doc_of, self.chunk_of = doc_of, chunk_of

    def __len__(self):
        return len(self.doc_of)

    def __getitem__(self, i: int) -> Tuple[int, int]:
        return int(self.doc_of[i]), int(self.chunk_of[i])


class BM25Retriever:
    """bm25_retriever.py:42-114 (langchain BaseRetriever upstream; the retrieval methods keep their names)."""

    def __init__(self, text_indexes: List[Tuple[int, int]], k: int, bm25: DeviceBM25, vocab: Dict[Hashable, int],
                 preprocess: Optional[Callable[[str], List[str]]] = None, remap: Optional[np.ndarray] = None):
        self.text_indexes = text_indexes  # flat index -> (doc_index, chunk_index)
        self.k = k
        self.bm25 = bm25
        self.vocab = vocab
        self._remap = remap  # vocabulary id -> this model's term id (-1: not in this corpus); None = identity
        self._preprocess = preprocess
        # concurrent single-query calls (bm25_retriever.py:102-104 runs them on executor threads) share passes;
        # one pass serves up to 512 queries, grouped by the n they ask for
        self._commits: Dict[int, _GroupCommit] = {}
        self._commits_lock = threading.Lock()

    @staticmethod
    def _get_text_index_gen(doc_records) -> Iterable[Tuple[int, object]]:
        # bm25_retriever.py:47-54
        for i, doc in enumerate(doc_records):
            if doc.text_index is not None:
                for item in doc.text_index:
                    yield i, item

    @staticmethod
    def has_index(document_records) -> bool:
        # bm25_retriever.py:56-62
        return sum(len(item.tokenized_text) for _, item in BM25Retriever._get_text_index_gen(document_records)) > 0

    @classmethod
    def from_doc_records(cls, doc_records, k: int = 4, device: int = 0,
                         preprocess: Optional[Callable[[str], List[str]]] = None) -> "BM25Retriever":
        # bm25_retriever.py:64-79.  The reference rebuilds BM25Okapi from scratch in every request (a Python loop
        # over every token of every document).  Here three things outlive the request (retrievers/_device_cache.py):
        # the model of a given tuple of documents (postings in HBM); every document's token-id arrays, so that a
        # NEW combination of known documents costs one concatenate + mir_bm25_create and no per-token Python; and
        # the process-wide str -> term-id vocabulary those arrays are written in (compacted per model to the ids
        # the corpus uses, in first-appearance order = the order of rank-bm25's dicts: mir_compact_term_ids).
        from ._device_cache import CACHE

        def build():
            docs = [(i, CACHE.get_or_build("bm25doc", -1, [doc.text_index], lambda d=doc: _doc_token_ids(d.text_index)))
                    for i, doc in enumerate(doc_records) if doc.text_index is not None]
            lens = np.concatenate([p[1] for _, p in docs]) if docs else np.zeros(0, np.int64)
            if int(lens.sum()) == 0:
                raise ValueError("Text index is empty.")
            doc_of = np.concatenate([np.full(len(p[0]), i, np.int64) for i, p in docs])
            chunk_of = np.concatenate([p[0] for _, p in docs])
            indptr = np.zeros(len(lens) + 1, np.int64)
            np.cumsum(lens, out=indptr[1:])
            ids = np.concatenate([p[2] for _, p in docs])
            # the model's per-term tables are sized by its vocabulary: compact the process-wide ids to this corpus
            remap = np.empty(int(ids.max()) + 1, np.int32)
            used = C.c_int32()
            nat.check(nat.lib.mir_compact_term_ids(nat.ptr(ids), len(ids), len(remap), nat.ptr(ids), nat.ptr(remap), C.byref(used)))
            bm25 = DeviceBM25.from_token_ids(indptr, ids, max(1, used.value), device=device)
            return (_FlatIndexTable(doc_of, chunk_of), bm25, remap), bm25.info()["hbm_bytes"]

        sources = [doc.text_index for doc in doc_records]  # None entries included: they shape the doc numbering
        text_indexes, bm25, remap = CACHE.get_or_build("bm25", device, sources, build)
        return cls(text_indexes=text_indexes, k=k, bm25=bm25, vocab=_VOCAB, preprocess=preprocess, remap=remap)

    def _ids(self, tokens: Sequence[Hashable]) -> List[int]:
        if self._remap is None:
            return [self.vocab.get(t, -1) for t in tokens]
        rm, n = self._remap, len(self._remap)
        out = []
        for t in tokens:
            g = self.vocab.get(t, -1)
            out.append(int(rm[g]) if 0 <= g < n else -1)
        return out

    def term_id(self, token: Hashable) -> int:
        """This model's id of a token, -1 if the corpus does not have it."""
        return self._ids([token])[0]

    def _get_top_n_indexes(self, query: List[str], n: int = 5) -> np.ndarray:
        # bm25_retriever.py:81-84
        with self._commits_lock:
            gc = self._commits.get(n)
            if gc is None:
                gc = self._commits[n] = _GroupCommit(lambda qs, n=n: self.bm25.search(qs, n), max_batch=512)
        idx, _, cnt = gc.submit(self._ids(query))
        return idx[: int(cnt)]

    def get_metadata_doc(self, index: int) -> Document:
        doc_index, chunk_index = self.text_indexes[index]
        return to_metadata_doc(doc_index, chunk_index, RetrievalType.TEXT)

    def _processed(self, query: str) -> List[str]:
        if self._preprocess is not None:
            return self._preprocess(query)
        from ..keywords_search import keywords_preprocess

        return keywords_preprocess(query)

    def _get_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        # bm25_retriever.py:92-97
        top_n = self._get_top_n_indexes(self._processed(query), self.k)
        return [self.get_metadata_doc(int(i)) for i in top_n]

    async def _aget_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        # bm25_retriever.py:99-104
        return await asyncio.get_running_loop().run_in_executor(None, self._get_relevant_documents, query)

    def get_relevant_documents_batch(self, queries: Sequence[str]) -> List[List[Document]]:
        idx, _, cnt = self.bm25.search([self._ids(self._processed(q)) for q in queries], self.k)
        return [[self.get_metadata_doc(int(i)) for i in idx[b, : cnt[b]]] for b in range(len(queries))]

    invoke = _get_relevant_documents

    @staticmethod
    async def build_index(chunks, stageio=None, preprocess: Optional[Callable[[str], List[str]]] = None) -> List[TextIndexItem]:
        # bm25_retriever.py:106-114 (runs on the indexing CPU pool upstream; tokenisation is host work)
        return await asyncio.get_running_loop().run_in_executor(None, _build_text_index_chunks, list(chunks), preprocess)
