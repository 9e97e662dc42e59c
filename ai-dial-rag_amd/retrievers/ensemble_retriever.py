"""Weighted reciprocal-rank fusion over member retrievers.

The reference uses langchain's ``EnsembleRetriever(retrievers, weights)``
(aidial_rag/retrieval_chain.py:239-245); this class keeps that constructor and
``invoke`` / ``_get_relevant_documents`` and fuses with ``mir_rrf_fuse``
(c = 60, documents keyed by ``page_content`` = "{doc_id}_{chunk_id}").
Members run concurrently on threads, as the async langchain path runs them with
``asyncio.gather``.
"""

import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence

import numpy as np

from .. import _native as nat
from ..index_record import Document


def weighted_reciprocal_rank(doc_lists: Sequence[Sequence[Document]], weights: Sequence[float], c: int = 60) -> List[Document]:
    if len(doc_lists) != len(weights):
        raise ValueError("Number of rank lists must be equal to the number of weights.")
    ids = {}
    first_doc = []
    flat = []
    ptr = [0]
    for docs in doc_lists:
        for d in docs:
            i = ids.setdefault(d.page_content, len(ids))
            if i == len(first_doc):
                first_doc.append(d)
            flat.append((i, 0))
        ptr.append(len(flat))
    keys = np.ascontiguousarray(np.array(flat, dtype=np.int64).reshape(-1, 2))
    lp = np.asarray(ptr, dtype=np.int32)
    w = np.asarray(weights, dtype=np.float64)
    out_keys = np.zeros((max(len(flat), 1), 2), np.int64)
    out_scores = np.zeros(max(len(flat), 1), np.float64)
    cnt = C.c_int32(0)
    nat.check(nat.lib.mir_rrf_fuse(nat.ptr(keys) if len(flat) else None, nat.ptr(lp), nat.ptr(w) if len(w) else None, len(doc_lists), c,
                                   nat.ptr(out_keys), nat.ptr(out_scores), C.byref(cnt)))
    return [first_doc[int(out_keys[r, 0])] for r in range(cnt.value)]


class EnsembleRetriever:
    def __init__(self, retrievers: Sequence, weights: Sequence[float] | None = None, c: int = 60):
        self.retrievers = list(retrievers)
        self.weights = list(weights) if weights is not None else [1.0 / len(self.retrievers)] * len(self.retrievers)
        self.c = c

    def _get_relevant_documents(self, query, *args, **kwargs) -> List[Document]:
        def run(r):
            return r.invoke(query) if hasattr(r, "invoke") else r._get_relevant_documents(query)

        if len(self.retrievers) > 1:
            with ThreadPoolExecutor(max_workers=len(self.retrievers)) as ex:
                lists = list(ex.map(run, self.retrievers))
        else:
            lists = [run(r) for r in self.retrievers]
        return weighted_reciprocal_rank(lists, self.weights, self.c)

    invoke = _get_relevant_documents
