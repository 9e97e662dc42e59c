"""Brute-force vector index, same surface as
aidial_rag/retrievers/embeddings_index.py:14-164.

``EmbeddingsIndex(retrieval_type, indexes, metric, limit).find(query)`` returns
the same ``List[Document]`` as the reference: the global stable ordering on
(distance, document position, row position) (embeddings_index.py:51-89).  The
rows of all ``DocIndex`` objects are flattened in that order into ONE device
index (``mir_index_create``); per query the GPU scans it once, re-scores the
surviving candidates in float64 and breaks ties on the flattened row - which is
exactly the reference's two-level stable argsort.

Added over the reference (it has no batch API): ``find_batch`` and
``search_arrays`` take B queries per pass; a single-query ``find`` is the B = 1
case of the same kernels and returns identical results.
"""

import ctypes as C
import threading
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import numpy.typing as npt

from .. import _native as nat
from ..index_record import Document, RetrievalType, to_metadata_doc
from ._group_commit import _GroupCommit
from .embeddings_metrics import ENUM_TO_METRIC, Metric  # noqa: F401  (re-exported like upstream)


class DocIndex:
    """embeddings_index.py:14-30."""

    chunk_ids: npt.NDArray[np.int64]
    embeddings: np.ndarray

    def __init__(self, chunk_ids: Optional[np.ndarray] = None, embeddings: Optional[np.ndarray] = None):
        self.chunk_ids = chunk_ids if chunk_ids is not None else np.array([], dtype=np.int64)
        self.embeddings = embeddings if embeddings is not None else np.array([], dtype=np.float32)


class DeviceRows:
    """Owner of one ``mir_rows`` handle: ONE document's rows (and chunk ids) resident in HBM.  Indexes over any
    set of documents are composed from these device-to-device (``DeviceIndex.from_rows``)."""

    def __init__(self, handle: C.c_void_p, n: int, d: int, device: int):
        self._h, self.n, self.d, self.device = handle, n, d, device

    @classmethod
    def from_host(cls, emb: np.ndarray, chunk_ids=None, device: int = 0):
        emb = np.asarray(emb)
        dtype = nat.DTYPE_F16 if emb.dtype == np.float16 else nat.DTYPE_F32
        emb = np.ascontiguousarray(emb, dtype=np.float16 if dtype == nat.DTYPE_F16 else np.float32)
        if emb.ndim != 2:
            raise ValueError(f"embeddings must be [n, d], got {emb.shape}")
        n, d = emb.shape
        ci = None if chunk_ids is None else np.ascontiguousarray(chunk_ids, dtype=np.int64)
        if ci is not None and len(ci) != n:
            raise ValueError(f"{len(ci)} chunk ids for {n} rows")
        h = C.c_void_p()
        nat.check(nat.lib.mir_rows_create(nat.ptr(emb), n, d, dtype, nat.ptr(ci), device, C.byref(h)))
        return cls(h, n, d, device)

    @property
    def handle(self):
        return self._h

    def hbm_bytes(self) -> int:
        b = C.c_int64(0)
        nat.check(nat.lib.mir_rows_info(self._h, None, None, None, None, C.byref(b)))
        return int(b.value)

    def close(self):
        if self._h:
            nat.lib.mir_rows_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceIndex:
    """Owner of one ``mir_index`` handle (a flattened shard resident in HBM)."""

    def __init__(self, handle: C.c_void_p, n: int, d: int, device: int):
        self._h = handle
        self.n, self.d, self.device = n, d, device

    @classmethod
    def from_host(cls, emb: np.ndarray, chunk_ids=None, doc_ids=None, device: int = 0, row_offset: int = 0):
        emb = np.asarray(emb)
        dtype = nat.DTYPE_F16 if emb.dtype == np.float16 else nat.DTYPE_F32  # float16 matrices upload at half the bytes
        emb = np.ascontiguousarray(emb, dtype=np.float16 if dtype == nat.DTYPE_F16 else np.float32)
        if emb.ndim != 2:
            raise ValueError(f"embeddings must be [n, d], got {emb.shape}")
        n, d = emb.shape
        ci = None if chunk_ids is None else np.ascontiguousarray(chunk_ids, dtype=np.int64)
        di = None if doc_ids is None else np.ascontiguousarray(doc_ids, dtype=np.int32)
        h = C.c_void_p()
        nat.check(nat.lib.mir_index_create(nat.ptr(emb), n, d, dtype, nat.ptr(ci), nat.ptr(di), device, row_offset, C.byref(h)))
        return cls(h, n, d, device)

    @classmethod
    def from_rows(cls, parts: Sequence["DeviceRows"], doc_ids: Optional[Sequence[int]] = None, device: int = 0,
                  row_offset: int = 0):
        """Concatenate row blocks device-to-device, in order; rows of ``parts[p]`` get doc id ``doc_ids[p]`` (default p)."""
        if not parts:
            raise ValueError("no row blocks")
        arr = (C.c_void_p * len(parts))(*[p.handle for p in parts])
        di = None if doc_ids is None else np.ascontiguousarray(doc_ids, dtype=np.int32)
        if di is not None and len(di) != len(parts):
            raise ValueError(f"{len(di)} doc ids for {len(parts)} row blocks")
        h = C.c_void_p()
        nat.check(nat.lib.mir_index_create_from_rows(arr, nat.ptr(di), len(parts), device, row_offset, C.byref(h)))
        return cls(h, sum(p.n for p in parts), parts[0].d, device)

    @classmethod
    def from_device_ptr(cls, emb_ptr: int, n: int, d: int, device: int, row_offset: int = 0, chunk_ids_ptr: int = 0,
                        doc_ids_ptr: int = 0, stream: int = 0, float16: bool = False):
        """Build from a float32 (or float16) [n, d] matrix already in HBM (e.g. ``tensor.data_ptr()``)."""
        h = C.c_void_p()
        nat.check(nat.lib.mir_index_create_from_device(emb_ptr or None, n, d, nat.DTYPE_F16 if float16 else nat.DTYPE_F32, chunk_ids_ptr or None,
                                                       doc_ids_ptr or None, device, row_offset, stream or None, C.byref(h)))
        return cls(h, n, d, device)

    @property
    def handle(self):
        return self._h

    def hbm_bytes(self) -> int:
        b = C.c_int64(0)
        nat.check(nat.lib.mir_index_info(self._h, None, None, None, None, C.byref(b)))
        return int(b.value)

    def search(self, queries: np.ndarray, k: int, metric) -> Tuple[np.ndarray, ...]:
        """-> (doc_ids[b,k] i32, chunk_ids[b,k] i64, rows[b,k] i64, dist[b,k] f64, count[b] i32, flags[b] i32)"""
        q = nat.as_f64_queries(queries, self.d)
        b = q.shape[0]
        code = nat.METRIC_CODES[Metric(metric).value]
        doc = np.zeros((b, k), np.int32)
        chunk = np.zeros((b, k), np.int64)
        row = np.zeros((b, k), np.int64)
        dist = np.zeros((b, k), np.float64)
        cnt = np.zeros(b, np.int32)
        flg = np.zeros(b, np.int32)
        nat.check(nat.lib.mir_index_search(self._h, nat.ptr(q), b, k, code, nat.ptr(doc), nat.ptr(chunk), nat.ptr(row),
                                           nat.ptr(dist), nat.ptr(cnt), nat.ptr(flg)))
        return doc, chunk, row, dist, cnt, flg

    def search_device(self, q_ptr: int, b: int, k: int, metric, out_row_ptr: int, out_dist_ptr: int, out_count_ptr: int,
                      out_flags_ptr: int = 0, out_doc_ptr: int = 0, out_chunk_ptr: int = 0, stream: int = 0) -> None:
        """Asynchronous search with every buffer in HBM (pointers as ints)."""
        code = nat.METRIC_CODES[Metric(metric).value]
        nat.check(nat.lib.mir_index_search_device(self._h, q_ptr, b, k, code, out_doc_ptr or None, out_chunk_ptr or None,
                                                  out_row_ptr or None, out_dist_ptr or None, out_count_ptr,
                                                  out_flags_ptr or None, stream or None))

    def profile(self, enable: bool) -> None:
        nat.check(nat.lib.mir_index_profile(self._h, 1 if enable else 0))

    def profile_read(self, reset: bool = True) -> Tuple[int, float]:
        """-> (scan launches, summed scan-kernel milliseconds) since the last reset."""
        n, ms = C.c_int64(0), C.c_double(0.0)
        nat.check(nat.lib.mir_index_profile_read(self._h, 1 if reset else 0, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def scan_stats(self, reset: bool = True) -> dict:
        """Counters of the sieve (large float32 shards) since the last reset: candidates per filter launch, queries
        answered, queries handed to the exact pass, candidates listed after the first launch, rows evaluated in float64."""
        out = np.zeros(8, np.int64)
        nat.check(nat.lib.mir_index_scan_stats(self._h, 1 if reset else 0, nat.ptr(out)))
        q = max(int(out[2] + out[3]), 1)
        return {"queries": int(out[2] + out[3]), "to_exact_pass": int(out[3]),
                "candidates_per_query_first_launch": round(float(out[0]) / q, 1),
                "candidates_per_query_second_launch": round(float(out[1]) / q, 1),
                "listed_per_query_after_first_launch": round(float(out[4]) / q, 1),
                "evaluated_in_float64_per_query": round(float(out[5]) / q, 1),
                "int8_first_stage": bool(out[6])}

    def metric_eval(self, query: np.ndarray, metric) -> np.ndarray:
        q = nat.as_f64_queries(query, self.d)[0]
        out = np.empty(self.n, np.float64)
        nat.check(nat.lib.mir_index_metric_eval(self._h, nat.ptr(q), nat.METRIC_CODES[Metric(metric).value], nat.ptr(out)))
        return out

    def close(self):
        if self._h:
            nat.lib.mir_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EmbeddingsIndex:
    """embeddings_index.py:33-89."""

    retrieval_type: RetrievalType
    doc_indexes: List[DocIndex]
    metric: str
    limit: int

    def __init__(self, retrieval_type: RetrievalType, indexes, metric: Metric = Metric.SQEUCLIDEAN_DIST,
                 limit: int = 1, device: int = 0, cache_sources: Optional[Sequence[object]] = None):
        """`indexes`: List[DocIndex] as in the reference, or a zero-argument callable that returns it (built
        only when the rows are actually needed).  `cache_sources`: the objects the rows come from, one per
        document; when given, the device index is shared across requests through retrievers/_device_cache.py
        instead of being flattened and uploaded again."""
        self.retrieval_type = retrieval_type
        self.metric = metric
        self.limit = limit
        self._indexes = indexes
        self.device = device
        self._dev: Optional[DeviceIndex] = None
        self._built = False
        self._lock = threading.Lock()
        self._cache_sources = tuple(cache_sources) if cache_sources is not None else None
        self._commit = _GroupCommit(lambda qs: self.search_arrays(np.stack(qs)), validate=self._check_query)

    @property
    def doc_indexes(self) -> List[DocIndex]:
        if callable(self._indexes):
            self._indexes = self._indexes()
        return self._indexes

    def _upload(self):
        """Flatten non-empty documents in order (embeddings_index.py:67-69) -> (DeviceIndex or None, HBM bytes).

        With ``cache_sources`` (one source object per document) every document's rows are a `DeviceRows` block
        cached on its own, so an index over a NEW combination of known documents uploads nothing: it is composed
        device-to-device.  Without, the documents are flattened on the host and uploaded as one matrix."""
        docs = [(i, doc) for i, doc in enumerate(self.doc_indexes) if len(doc.embeddings) > 0]
        if not docs:
            return None, 0
        srcs = self._cache_sources
        n_docs = len(self.doc_indexes)
        if srcs is not None and len(srcs) >= n_docs and len(srcs) % n_docs == 0:
            from ._device_cache import CACHE

            m = len(srcs) // n_docs  # source objects per document (by-page indexes have two: chunks and page rows)

            def block(doc):
                def build():
                    r = DeviceRows.from_host(np.asarray(doc.embeddings), np.asarray(doc.chunk_ids, dtype=np.int64), self.device)
                    return r, r.hbm_bytes()

                return build

            parts = [CACHE.get_or_build("rows", self.device, srcs[i * m:(i + 1) * m], block(doc)) for i, doc in docs]
            dev = DeviceIndex.from_rows(parts, [i for i, _ in docs], self.device)
            return dev, dev.hbm_bytes()
        embs = [np.asarray(doc.embeddings, dtype=np.float32) for _, doc in docs]
        chunk_ids = [np.asarray(doc.chunk_ids, dtype=np.int64) for _, doc in docs]
        doc_ids = [np.full(len(e), i, dtype=np.int32) for (i, _), e in zip(docs, embs)]
        dev = DeviceIndex.from_host(np.concatenate(embs), np.concatenate(chunk_ids), np.concatenate(doc_ids), self.device)
        return dev, dev.hbm_bytes()

    def _device_index(self) -> Optional[DeviceIndex]:
        with self._lock:
            if not self._built:
                if self._cache_sources is not None:
                    from ._device_cache import CACHE

                    self._dev = CACHE.get_or_build("vector", self.device, self._cache_sources, self._upload)
                else:
                    self._dev = self._upload()[0]
                self._built = True
            return self._dev

    def _check_query(self, query) -> np.ndarray:
        """Shape / dtype of ONE query, checked in the caller's thread before it joins a shared pass: a malformed
        query raises for its own caller only."""
        q = np.asarray(query, dtype=np.float64)
        if q.ndim != 1:
            raise ValueError(f"query must be one vector, got shape {q.shape}")
        dev = self._device_index()
        if dev is not None and q.shape[0] != dev.d:
            raise ValueError(f"query shape {q.shape} does not match index dimension {dev.d}")
        return q

    def search_arrays(self, queries: np.ndarray):
        """B queries -> (doc_ids, chunk_ids, dist, count, flags) arrays, best first.  flags: 0, or
        nat.FLAG_EXACT_PASS for a query the exact pass answered; results are the reference's either way."""
        Metric(self.metric)  # unknown metric -> ValueError, as embeddings_index.py:54
        dev = self._device_index()
        if dev is None:
            q = np.atleast_2d(np.asarray(queries))
            b = q.shape[0]
            z = np.zeros((b, 0))
            return z.astype(np.int32), z.astype(np.int64), z, np.zeros(b, np.int32), np.zeros(b, np.int32)
        doc, chunk, _row, dist, cnt, flg = dev.search(queries, self.limit, self.metric)
        return doc, chunk, dist, cnt, flg

    def find_batch(self, queries: np.ndarray) -> List[List[Document]]:
        doc, chunk, _dist, cnt, _ = self.search_arrays(queries)
        return [
            [to_metadata_doc(int(doc[b, j]), int(chunk[b, j]), retrieval_type=self.retrieval_type) for j in range(int(cnt[b]))]
            for b in range(len(cnt))
        ]

    def find(self, query: np.ndarray) -> List[Document]:
        """One query (embeddings_index.py:62-89).  Concurrent callers share search passes (`_GroupCommit`);
        the result of a query does not depend on what else rides in its pass."""
        Metric(self.metric)
        doc, chunk, _dist, cnt, _flags = self._commit.submit(query)
        return [to_metadata_doc(int(doc[j]), int(chunk[j]), retrieval_type=self.retrieval_type) for j in range(int(cnt))]


def _get_page_index(chunk) -> int:
    # embeddings_index.py:92-94: page numbers are 1-based
    return chunk.metadata["page_number"] - 1


def _item_rows(item) -> np.ndarray:
    """Rows of one MultiEmbeddings item: an ``.embeddings`` attribute (docarray ItemEmbeddings) or a bare array."""
    return np.asarray(getattr(item, "embeddings", item))


def create_index_by_page(chunks: Sequence, pages_embeddings: Optional[Sequence]) -> DocIndex:
    """embeddings_index.py:101-118."""
    if pages_embeddings is None:
        return DocIndex()
    ids: List[int] = []
    rows: List[np.ndarray] = []
    for i, chunk in enumerate(chunks):
        emb = _item_rows(pages_embeddings[_get_page_index(chunk)])
        ids.extend([i] * len(emb))
        rows.extend(emb)
    return DocIndex(chunk_ids=np.array(ids, dtype=np.int64), embeddings=np.array(rows, dtype=np.float32))


def create_index_by_chunk(chunks_embeddings: Optional[Sequence]) -> DocIndex:
    """embeddings_index.py:121-136."""
    if chunks_embeddings is None:
        return DocIndex()
    ids: List[int] = []
    rows: List[np.ndarray] = []
    for i, item in enumerate(chunks_embeddings):
        emb = _item_rows(item)
        ids.extend([i] * len(emb))
        rows.extend(emb)
    return DocIndex(chunk_ids=np.array(ids, dtype=np.int64), embeddings=np.array(rows))


class ItemEmbeddings:
    """document_record.py:34-39: the embeddings of one chunk or page, [m, d] float32."""

    __slots__ = ("embeddings",)

    def __init__(self, embeddings: np.ndarray):
        self.embeddings = embeddings


def pack_multi_embeddings(indexes: List[int], embeddings: Iterable[np.ndarray], number_of_pages: int) -> List[ItemEmbeddings]:
    """embeddings_index.py:139-153."""
    pages: List[list] = [[] for _ in range(number_of_pages)]
    for page_index, e in zip(indexes, embeddings):
        pages[page_index].append(e)
    return [ItemEmbeddings(np.array(p, dtype=np.float32)) for p in pages]


def pack_simple_embeddings(embeddings: Iterable[np.ndarray]) -> List[ItemEmbeddings]:
    """embeddings_index.py:156-164."""
    return [ItemEmbeddings(np.array([e], dtype=np.float32)) for e in embeddings]
