#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container, where the upstream checkout is mounted
read-only at /root/reference.  It imports the reference's
``aidial_rag/retrievers/embeddings_metrics.py`` (the one hot-path module whose
imports - enum, numpy, torch - are all present; python 3.10 lacks
``enum.StrEnum``, so the 3.11 stdlib class is back-filled first) and records
its outputs on committed inputs.  Nothing of the reference travels: the
fixtures hold inputs and expected outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures written:
  metrics_known.json   every (query, docs) case of the reference's
                       tests/test_embeddings_metrics.py:6-201 with the values
                       the reference functions return (and the values those
                       tests assert)
  metrics_random.npz   seeded unit-norm f32[512,384] docs + 8 queries and a
                       d=1024 fp16-rounded non-normalised set; outputs of all
                       four reference metrics for float32 AND float64 queries
  metrics_ties.npz     engineered exact ties / 1-ulp near ties / duplicate of
                       the query (negative squared distance -> NaN euclid)
The index-level cases of tests/test_embeddings_index.py:11-94 are data only
(tiny arrays and expected id pairs) and live in index_cases.json, written by
hand from the test text; this script re-checks them with the oracle.
"""

import enum
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/aidial_rag/retrievers/embeddings_metrics.py"


def load_reference_metrics():
    if not hasattr(enum, "StrEnum"):  # python < 3.11

        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)

        enum.StrEnum = StrEnum
    spec = importlib.util.spec_from_file_location("_ref_embeddings_metrics", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


# (metric, query, docs, value asserted by the reference test) -- inputs are the
# literal arrays of tests/test_embeddings_metrics.py, in file order.
S2, S5, S22 = 2**0.5, 5**0.5, 22**0.5
KNOWN = [
    ("cosine_sim", [1.0, 0, 0, 0], [[1.0, 0, 0, 0], [0, 1.0, 0, 0]], [-1.0, 0.0]),
    ("cosine_sim", [-1.0, 0, 0, 0], [[1.0, 0, 0, 0], [0, 1.0, 0, 0]], [1.0, 0.0]),
    ("cosine_sim", [2.0, 0, 0, 0], [[1.0, 0, 0, 0], [0, 1.0, 0, 0]], [-1.0, 0.0]),
    ("cosine_sim", [0.0, 0, 0, 0], [[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0]], [0.0, 0.0, 0.0]),
    ("cosine_sim", [1.0, 0, 0, 0], [[2.0, 0, 0, 0], [0, 2.0, 0, 0], [0, 0, 0, 0.0]], [-1.0, 0.0, 0.0]),
    ("inner_product", [1, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [-1.0, 0.0]),
    ("inner_product", [-1, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [1.0, 0.0]),
    ("inner_product", [2, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [-2.0, 0.0]),
    ("inner_product", [0, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0]], [0.0, 0.0, 0.0]),
    ("inner_product", [1, 0, 0, 0], [[2, 0, 0, 0], [0, 2, 0, 0], [0, 0, 0, 0]], [-2.0, 0.0, 0.0]),
    ("euclidean_dist", [1, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [0.0, S2]),
    ("euclidean_dist", [-1, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [2.0, S2]),
    ("euclidean_dist", [2, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [1.0, S5]),
    ("euclidean_dist", [1, 0, 0, 0], [[2, 0, 0, 0], [3, 3, 3, 0], [0, 0, 0, 0]], [1.0, S22, 1.0]),
    ("sqeuclidean_dist", [1, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [0.0, 2.0]),
    ("sqeuclidean_dist", [-1, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [4.0, 2.0]),
    ("sqeuclidean_dist", [2, 0, 0, 0], [[1, 0, 0, 0], [0, 1, 0, 0]], [1.0, 5.0]),
    ("sqeuclidean_dist", [1, 0, 0, 0], [[2, 0, 0, 0], [3, 3, 3, 0], [0, 0, 0, 0]], [1.0, 22.0, 1.0]),
    ("sqeuclidean_dist", [0, 0, 0, 0], [[1, 1, 1, 1], [2, 2, 2, 2]], [4.0, 16.0]),
]
# the two equivalence tests (:96-121, :188-201): inputs only
EQUIV_DOCS = [[1, 0, 0, 0], [0, 1, 0, 0], [2, 0, 0, 0], [3, 3, 3, 0], [0, 0, 0, 0]]
EQUIV_QUERY = [1, 2, 3, 4]


def unit_rows(x):
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)


def main():
    ref = load_reference_metrics()
    fn = {str(k.value): f for k, f in ref.ENUM_TO_METRIC.items()}
    assert sorted(fn) == ["cosine_sim", "euclidean_dist", "inner_product", "sqeuclidean_dist"]

    # ---- 1. known-answer cases ------------------------------------------
    known = []
    for metric, q, d, asserted in KNOWN:
        qa = np.array(q) if any(isinstance(v, float) for v in q) else np.array(q)
        da = np.array(d)
        out = fn[metric](qa, da)
        np.testing.assert_allclose(out, np.array(asserted))  # what the reference test asserts
        known.append(
            {
                "metric": metric,
                "query": q,
                "docs": d,
                "query_dtype": str(qa.dtype),
                "docs_dtype": str(da.dtype),
                "asserted": asserted,
                "reference_out": [float(v) for v in out],
                "reference_dtype": str(out.dtype),
            }
        )
    with np.errstate(invalid="ignore"):
        # 0/0 row: the reference test normalises an all-zero row -> NaN row
        eq_docs = np.array(EQUIV_DOCS) / np.linalg.norm(np.array(EQUIV_DOCS), axis=-1, keepdims=True)
    eq_q = np.array(EQUIV_QUERY) / np.linalg.norm(np.array(EQUIV_QUERY))
    equiv = {
        "docs_raw": EQUIV_DOCS,
        "query_raw": EQUIV_QUERY,
        "cosine_norm": [float(v) for v in fn["cosine_sim"](eq_q, eq_docs)],
        "inner_norm": [float(v) for v in fn["inner_product"](eq_q, eq_docs)],
        "euclid_raw": [float(v) for v in fn["euclidean_dist"](np.array(EQUIV_QUERY), np.array(EQUIV_DOCS))],
        "sqeuclid_raw": [float(v) for v in fn["sqeuclidean_dist"](np.array(EQUIV_QUERY), np.array(EQUIV_DOCS))],
    }
    with open(os.path.join(HERE, "metrics_known.json"), "w") as f:
        json.dump({"cases": known, "equivalence": equiv}, f, indent=1)

    # ---- 2. seeded random sets -------------------------------------------
    rng = np.random.default_rng(20250905)
    docs384 = unit_rows(rng.standard_normal((512, 384)))
    q384 = unit_rows(rng.standard_normal((8, 384)))
    docs1024 = rng.standard_normal((256, 1024)).astype(np.float16).astype(np.float32)  # fp16-representable, not normalised
    q1024 = rng.standard_normal((4, 1024)).astype(np.float32)
    out = {"docs384": docs384, "q384": q384, "docs1024": docs1024, "q1024": q1024}
    for tag, docs, qs in (("384", docs384, q384), ("1024", docs1024, q1024)):
        for name, f in fn.items():
            out[f"{name}_{tag}_f32"] = np.stack([f(q, docs) for q in qs])
            out[f"{name}_{tag}_f64"] = np.stack([f(q.astype(np.float64), docs) for q in qs])
            assert out[f"{name}_{tag}_f32"].dtype == np.float32 and out[f"{name}_{tag}_f64"].dtype == np.float64
    np.savez_compressed(os.path.join(HERE, "metrics_random.npz"), **out)

    # ---- 3. ties, near ties, self-match ----------------------------------
    base = unit_rows(rng.standard_normal((64, 384)))
    docs = base.copy()
    docs[10] = docs[3]  # exact duplicates -> exact ties, lower row must win
    docs[40] = docs[3]
    docs[21] = docs[20]
    docs[21, 0] = np.nextafter(docs[21, 0], np.float32(2.0))  # 1-ulp neighbour of row 20
    queries = np.stack([docs[3], docs[20], unit_rows(rng.standard_normal((1, 384)))[0], np.zeros(384, np.float32)])
    tout = {"docs": docs, "queries": queries}
    for name, f in fn.items():
        with np.errstate(invalid="ignore"):
            tout[f"{name}_f64"] = np.stack([f(q.astype(np.float64), docs) for q in queries])
            tout[f"{name}_f32"] = np.stack([f(q, docs) for q in queries])
    np.savez_compressed(os.path.join(HERE, "metrics_ties.npz"), **tout)

    # ---- 4. re-check the hand-written index cases with the oracle ---------
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import embeddings_index as oi

    with open(os.path.join(HERE, "index_cases.json")) as f:
        cases = json.load(f)
    docs_by_name = {
        k: oi.DocIndex(np.array(v["chunk_ids"], dtype=np.int64), np.array(v["embeddings"], dtype=np.float32))
        for k, v in cases["docs"].items()
    }
    n = 0
    for case in cases["cases"]:
        for metric in cases["metrics"]:
            got, _ = oi.find(
                np.array(case["query"]), [docs_by_name[d] for d in case["doc_order"]], metric, case["limit"]
            )
            assert got == [tuple(p) for p in case["expected"]], (case, metric, got)
            n += 1
    print(f"golden fixtures written; {len(known)} known-answer cases, {n} index cases re-checked")


if __name__ == "__main__":
    main()
