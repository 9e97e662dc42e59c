"""Oracle: brute-force per-document top-k and the cross-document merge.

Restates aidial_rag/retrievers/embeddings_index.py:14-89 (DocIndex,
EmbeddingsIndex.find_in_doc / .find) and the index builders :92-164.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Results are returned as ``(doc_id, chunk_id)`` integer pairs: the reference
wraps exactly these two integers into ``to_metadata_doc`` (index_record.py:
29-38), so the pairs are the parity currency.
"""

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .embeddings_metrics import ENUM_TO_METRIC, Metric


class DocIndex:
    """embeddings_index.py:14-30: chunk_ids int64[M], embeddings [M, d]."""

    def __init__(self, chunk_ids: Optional[np.ndarray] = None, embeddings: Optional[np.ndarray] = None):
        self.chunk_ids = chunk_ids if chunk_ids is not None else np.array([], dtype=np.int64)
        self.embeddings = embeddings if embeddings is not None else np.array([], dtype=np.float32)


def find_in_doc(query: np.ndarray, doc: DocIndex, metric, limit: int):
    """embeddings_index.py:51-60: metric, then STABLE argsort, first `limit`."""
    dist = ENUM_TO_METRIC[Metric(metric)](query, doc.embeddings)
    top = np.argsort(dist, kind="stable")[:limit]
    return doc.chunk_ids[top], dist[top]


def find(
    query: np.ndarray, doc_indexes: Sequence[DocIndex], metric=Metric.SQEUCLIDEAN_DIST, limit: int = 1
) -> Tuple[List[Tuple[int, int]], np.ndarray]:
    """embeddings_index.py:62-89.

    Per-document top-`limit`, concatenated in document order, then a second
    stable argsort.  Empty documents are skipped (:68) but keep their position
    in the doc_id numbering.  No de-duplication.  Returns the (doc_id,
    chunk_id) pairs and their distances.
    """
    doc_ids = np.array([], dtype=np.int64)
    chunk_ids = np.array([], dtype=np.int64)
    dists = np.array([], dtype=np.float32)
    for i, doc in enumerate(doc_indexes):
        if len(doc.embeddings) == 0:
            continue
        c, d = find_in_doc(query, doc, metric, limit)
        doc_ids = np.concatenate((doc_ids, np.full(len(c), i, dtype=np.int64)))
        chunk_ids = np.concatenate((chunk_ids, c))
        dists = np.concatenate((dists, d))
    top = np.argsort(dists, kind="stable")[:limit]
    pairs = [(int(a), int(b)) for a, b in zip(doc_ids[top], chunk_ids[top])]
    return pairs, dists[top]


def find_flat(
    query: np.ndarray, embeddings: np.ndarray, metric=Metric.SQEUCLIDEAN_DIST, limit: int = 1
) -> Tuple[np.ndarray, np.ndarray]:
    """The same search over ONE flattened matrix: row indices + distances.

    `find` over documents is equivalent to one global stable sort on
    (distance, doc_index, row_index); with all rows in one
    DocIndex that is argsort(stable) of the distances.  Used for large-N cases
    where the flat row index is the id.
    """
    dist = ENUM_TO_METRIC[Metric(metric)](query, embeddings)
    top = np.argsort(dist, kind="stable")[:limit]
    return top.astype(np.int64), dist[top]


# ---- index builders (embeddings_index.py:92-164) ---------------------------
# A "MultiEmbeddings" is represented here as a plain list of [m_i, d] arrays
# (the docarray DocList[ItemEmbeddings] wrapper carries nothing else).


def create_index_by_chunk(chunks_embeddings: Optional[Sequence[np.ndarray]]) -> DocIndex:
    """embeddings_index.py:121-136: row j of chunk i gets chunk_id i."""
    if chunks_embeddings is None:
        return DocIndex()
    ids: List[int] = []
    rows: List[np.ndarray] = []
    for i, emb in enumerate(chunks_embeddings):
        ids.extend([i] * len(emb))
        rows.extend(emb)
    return DocIndex(np.array(ids, dtype=np.int64), np.array(rows))


def create_index_by_page(
    chunk_page_numbers: Sequence[int], pages_embeddings: Optional[Sequence[np.ndarray]]
) -> DocIndex:
    """embeddings_index.py:92-118: every chunk repeats the rows of its page.

    `chunk_page_numbers[i]` is chunk i's 1-based ``metadata["page_number"]``.
    """
    if pages_embeddings is None:
        return DocIndex()
    ids: List[int] = []
    rows: List[np.ndarray] = []
    for i, page in enumerate(chunk_page_numbers):
        emb = pages_embeddings[page - 1]
        ids.extend([i] * len(emb))
        rows.extend(emb)
    return DocIndex(np.array(ids, dtype=np.int64), np.array(rows, dtype=np.float32))


def pack_simple_embeddings(embeddings) -> List[np.ndarray]:
    """embeddings_index.py:156-164: one [1, d] float32 array per item."""
    return [np.array([e], dtype=np.float32) for e in embeddings]


def pack_multi_embeddings(indexes, embeddings, number_of_pages: int) -> List[np.ndarray]:
    """embeddings_index.py:139-153: group rows by page index."""
    pages: List[list] = [[] for _ in range(number_of_pages)]
    for page_index, e in zip(indexes, embeddings):
        pages[page_index].append(e)
    return [np.array(p, dtype=np.float32) for p in pages]
