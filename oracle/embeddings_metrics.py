"""Oracle: the four "smaller is better" vector metrics.

Restates aidial_rag/retrievers/embeddings_metrics.py:7-58.  TEST
INFRASTRUCTURE ONLY (see oracle/__init__.py).

The result dtype is the numpy promotion of the inputs, exactly as in the
reference: the live path passes a float64 query (semantic_retriever.py:49,53
builds ``np.array(List[float])``) against float32 docs, so scores are float64.
"""

from enum import Enum

import numpy as np


class Metric(str, Enum):
    # embeddings_metrics.py:7-11 (StrEnum there; str+Enum has the same values
    # and the same ``Metric("bad") -> ValueError`` behaviour on python 3.10)
    COSINE_SIM = "cosine_sim"
    EUCLIDEAN_DIST = "euclidean_dist"
    SQEUCLIDEAN_DIST = "sqeuclidean_dist"
    INNER_PRODUCT = "inner_product"

    def __str__(self) -> str:  # StrEnum.__str__
        return str(self.value)


COSINE_EPS = 1e-8  # torch.nn.functional.cosine_similarity default eps


def metric_inner_product(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    # embeddings_metrics.py:14-20: -np.inner(query, docs)
    return -np.inner(query, docs)


def metric_cosine_sim(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    # embeddings_metrics.py:23-31: -torch.cosine_similarity(docs, query).
    # torch 2.x (ATen/native/Distance.cpp) normalises EACH operand in its own
    # dtype first - L2 norm clamped at eps, then x / norm - and only the final
    # product promotes: sum((docs/|docs|) * (query/|query|)).  With float32 docs
    # and a float64 query the unit doc rows are therefore rounded to float32
    # before the float64 multiply-sum.  Zero vectors give 0, not NaN (pinned by
    # tests/test_embeddings_metrics.py:32-39).  Integer inputs are promoted to
    # the common floating dtype first, as torch does.
    dt = np.result_type(query.dtype, docs.dtype)
    if not np.issubdtype(dt, np.floating):
        dt = np.dtype(np.float64)
    q = query if np.issubdtype(query.dtype, np.floating) else query.astype(dt)
    d = docs if np.issubdtype(docs.dtype, np.floating) else docs.astype(dt)
    qn = np.maximum(np.sqrt(np.sum(q * q)), q.dtype.type(COSINE_EPS))
    dn = np.maximum(np.sqrt(np.sum(d * d, axis=-1, keepdims=True)), d.dtype.type(COSINE_EPS))
    return -np.sum((d / dn) * (q / qn), axis=-1)


def metric_sqeuclidean_dist(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    # embeddings_metrics.py:34-43.  Deliberately the expanded form
    # doc_sq - 2*dot + query_sq, NOT sum((docs-query)**2): doc_sq is summed in
    # the docs' own dtype (float32 pairwise sum), the dot is promoted.
    doc_sq = np.sum(docs**2, axis=1)
    query_sq = np.sum(query**2)
    query_dot = np.dot(docs, query)
    return doc_sq - 2 * query_dot + query_sq


def metric_euclidean_dist(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    # embeddings_metrics.py:46-50.  A slightly negative squared distance
    # (rounding, near-identical vectors) becomes NaN here, as upstream.
    with np.errstate(invalid="ignore"):
        return np.sqrt(metric_sqeuclidean_dist(query, docs))


ENUM_TO_METRIC = {
    Metric.COSINE_SIM: metric_cosine_sim,
    Metric.EUCLIDEAN_DIST: metric_euclidean_dist,
    Metric.SQEUCLIDEAN_DIST: metric_sqeuclidean_dist,
    Metric.INNER_PRODUCT: metric_inner_product,
}
assert len(ENUM_TO_METRIC) == len(Metric)
