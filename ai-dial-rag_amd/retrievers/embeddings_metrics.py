"""Vector metrics, same surface as aidial_rag/retrievers/embeddings_metrics.py:7-58.

``ENUM_TO_METRIC[Metric(m)](query, docs) -> np.ndarray`` with the reference's
sign convention (smaller is better) and result dtype (numpy promotion of the
inputs: float64 on the live path where the query is float64).  The arithmetic
runs on the GPU (``mir_metric_eval``: one wave per row, float64, the
reference's formulas); docs are held as float32, the dtype the product stores.
"""

import ctypes as C
from enum import Enum

import numpy as np

from .. import _native as nat


class Metric(str, Enum):  # embeddings_metrics.py:7-11
    COSINE_SIM = "cosine_sim"
    EUCLIDEAN_DIST = "euclidean_dist"
    SQEUCLIDEAN_DIST = "sqeuclidean_dist"
    INNER_PRODUCT = "inner_product"

    def __str__(self) -> str:
        return str(self.value)


def _result_dtype(metric: "Metric", query: np.ndarray, docs: np.ndarray):
    """numpy promotion of the inputs, as the reference's expressions give it:
    all-integer inputs stay integer for inner product / squared distance
    (tests/test_embeddings_metrics.py passes int arrays), sqrt and cosine are floating."""
    dt = np.result_type(query.dtype, docs.dtype)
    if np.issubdtype(dt, np.floating):
        return dt
    if metric in (Metric.INNER_PRODUCT, Metric.SQEUCLIDEAN_DIST) and np.issubdtype(dt, np.integer):
        return dt
    return np.dtype(np.float64)


def _eval(metric: Metric, query: np.ndarray, docs: np.ndarray, device: int = 0) -> np.ndarray:
    query = np.asarray(query)
    docs = np.asarray(docs)
    if docs.ndim != 2:
        raise ValueError(f"docs must be [n, d], got shape {docs.shape}")
    n, d = docs.shape
    q64 = nat.as_f64_queries(query, d)[0]
    out = np.empty(n, dtype=np.float64)
    if n:
        d32 = np.ascontiguousarray(docs, dtype=np.float32)
        nat.check(
            nat.lib.mir_metric_eval(nat.ptr(d32), n, d, nat.DTYPE_F32, nat.ptr(q64), nat.METRIC_CODES[metric.value], device, nat.ptr(out))
        )
    dt = _result_dtype(metric, query, docs)
    if np.issubdtype(dt, np.integer):
        return np.rint(out).astype(dt)
    return out.astype(dt, copy=False)


def _metric_for_inner_product(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    return _eval(Metric.INNER_PRODUCT, query, docs)


def _metric_for_cosine_sim(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    return _eval(Metric.COSINE_SIM, query, docs)


def _metric_for_sqeuclidean_dist(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    return _eval(Metric.SQEUCLIDEAN_DIST, query, docs)


def _metric_for_euclidean_dist(query: np.ndarray, docs: np.ndarray) -> np.ndarray:
    return _eval(Metric.EUCLIDEAN_DIST, query, docs)


ENUM_TO_METRIC = {
    Metric.COSINE_SIM: _metric_for_cosine_sim,
    Metric.EUCLIDEAN_DIST: _metric_for_euclidean_dist,
    Metric.SQEUCLIDEAN_DIST: _metric_for_sqeuclidean_dist,
    Metric.INNER_PRODUCT: _metric_for_inner_product,
}

assert len(ENUM_TO_METRIC) == len(Metric)
