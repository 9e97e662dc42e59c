"""The oracle against the golden vectors recorded from the reference
(tests/golden/make_golden.py) and the reference's own known-answer cases
(epam/ai-dial-rag tests/test_embeddings_metrics.py, tests/test_embeddings_index.py)."""

import json
import os

import numpy as np
import pytest

from oracle import embeddings_index as oi
from oracle.embeddings_metrics import ENUM_TO_METRIC, Metric

METRICS = [m.value for m in Metric]


def test_metric_enum_values_and_errors():
    assert sorted(METRICS) == ["cosine_sim", "euclidean_dist", "inner_product", "sqeuclidean_dist"]
    with pytest.raises(ValueError):
        Metric("bad")


def test_known_answer_cases(golden_dir):
    data = json.load(open(os.path.join(golden_dir, "metrics_known.json")))
    assert len(data["cases"]) == 19
    for c in data["cases"]:
        q = np.array(c["query"], dtype=c["query_dtype"])
        d = np.array(c["docs"], dtype=c["docs_dtype"])
        out = ENUM_TO_METRIC[Metric(c["metric"])](q, d)
        np.testing.assert_allclose(out, np.array(c["asserted"]))  # the reference test's assertion
        np.testing.assert_array_equal(out, np.array(c["reference_out"]))  # what the reference returned
        assert str(out.dtype) == c["reference_dtype"]


def test_equivalence_cases(golden_dir):
    eq = json.load(open(os.path.join(golden_dir, "metrics_known.json")))["equivalence"]
    docs = np.array(eq["docs_raw"])
    q = np.array(eq["query_raw"])
    np.testing.assert_allclose(ENUM_TO_METRIC[Metric.EUCLIDEAN_DIST](q, docs) ** 2, ENUM_TO_METRIC[Metric.SQEUCLIDEAN_DIST](q, docs))
    np.testing.assert_allclose(ENUM_TO_METRIC[Metric.EUCLIDEAN_DIST](q, docs), eq["euclid_raw"])
    np.testing.assert_allclose(ENUM_TO_METRIC[Metric.SQEUCLIDEAN_DIST](q, docs), eq["sqeuclid_raw"])
    with np.errstate(invalid="ignore"):
        nd = docs / np.linalg.norm(docs, axis=-1, keepdims=True)
    nq = q / np.linalg.norm(q)
    np.testing.assert_allclose(ENUM_TO_METRIC[Metric.COSINE_SIM](nq, nd), eq["cosine_norm"])
    np.testing.assert_allclose(ENUM_TO_METRIC[Metric.INNER_PRODUCT](nq, nd), eq["inner_norm"])


@pytest.mark.parametrize("tag", ["384", "1024"])
@pytest.mark.parametrize("metric", METRICS)
def test_random_sets_match_reference_outputs(golden_dir, tag, metric):
    z = np.load(os.path.join(golden_dir, "metrics_random.npz"))
    docs, qs = z[f"docs{tag}"], z[f"q{tag}"]
    f = ENUM_TO_METRIC[Metric(metric)]
    for i, q in enumerate(qs):
        o64 = f(q.astype(np.float64), docs)
        o32 = f(q, docs)
        assert o64.dtype == np.float64 and o32.dtype == np.float32
        scale = 1.0 if tag == "384" else 1024.0
        # float64 path: same libraries, differences only from summation order;
        # cosine normalises the float32 docs in float32 (torch), so the unit rows
        # carry a 1-ulp(f32) dependence on the norm's summation order (~1e-8)
        atol64 = 3e-8 if metric == "cosine_sim" else 1e-12 * scale
        np.testing.assert_allclose(o64, z[f"{metric}_{tag}_f64"][i], rtol=0, atol=atol64)
        np.testing.assert_allclose(o32, z[f"{metric}_{tag}_f32"][i], rtol=0, atol=2e-6 * scale)
        # ranking parity on the float64 (live) path
        np.testing.assert_array_equal(
            np.argsort(o64, kind="stable")[:10], np.argsort(z[f"{metric}_{tag}_f64"][i], kind="stable")[:10]
        )


@pytest.mark.parametrize("metric", METRICS)
def test_ties_and_self_match(golden_dir, metric):
    z = np.load(os.path.join(golden_dir, "metrics_ties.npz"))
    docs, queries = z["docs"], z["queries"]
    f = ENUM_TO_METRIC[Metric(metric)]
    for i, q in enumerate(queries):
        with np.errstate(invalid="ignore"):
            o = f(q.astype(np.float64), docs)
        ref = z[f"{metric}_f64"][i]
        np.testing.assert_array_equal(np.isnan(o), np.isnan(ref))
        np.testing.assert_allclose(o, ref, rtol=0, atol=1.5e-7 if metric == "cosine_sim" else 1e-12, equal_nan=True)
        if metric != "cosine_sim":  # 1e-8 noise (above) may reorder unrelated rows
            np.testing.assert_array_equal(np.argsort(o, kind="stable"), np.argsort(ref, kind="stable"))
    # duplicates of row 3 tie exactly and resolve to the lower row
    o = f(queries[0].astype(np.float64), docs)
    order = [int(v) for v in np.argsort(o, kind="stable")]
    if metric == "euclidean_dist":
        # upstream quirk kept on purpose: the self-match's squared distance
        # rounds slightly negative, sqrt gives NaN, and NaN sorts LAST
        assert np.isnan(o[[3, 10, 40]]).all() and order[-3:] == [3, 10, 40]
    else:
        assert o[3] == o[10] == o[40]
        assert order[:3] == [3, 10, 40]


def test_index_cases(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "index_cases.json")))
    docs = {
        k: oi.DocIndex(np.array(v["chunk_ids"], dtype=np.int64), np.array(v["embeddings"], dtype=np.float32))
        for k, v in cases["docs"].items()
    }
    for case in cases["cases"]:
        for metric in cases["metrics"]:
            got, _ = oi.find(np.array(case["query"]), [docs[d] for d in case["doc_order"]], metric, case["limit"])
            assert got == [tuple(p) for p in case["expected"]], (case["name"], metric)


def test_find_equals_global_stable_sort():
    rng = np.random.default_rng(5)
    parts = [rng.standard_normal((m, 16)).astype(np.float32) for m in (5, 0, 17, 1, 9)]
    parts[2][4] = parts[0][1]  # cross-document exact tie
    docs = [oi.DocIndex(np.arange(len(p), dtype=np.int64), p) if len(p) else oi.DocIndex() for p in parts]
    flat = np.concatenate([p for p in parts if len(p)])
    owner = np.concatenate([np.full(len(p), i) for i, p in enumerate(parts)])
    local = np.concatenate([np.arange(len(p)) for p in parts])
    for metric in METRICS:
        for limit in (1, 3, 7, 50):
            q = parts[0][1].astype(np.float64)
            pairs, dist = oi.find(q, docs, metric, limit)
            rows, fdist = oi.find_flat(q, flat, metric, limit)
            assert pairs == [(int(owner[r]), int(local[r])) for r in rows]
            np.testing.assert_array_equal(dist, fdist)


def test_index_builders():
    embs = [np.ones((1, 4), np.float32) * i for i in range(3)]
    idx = oi.create_index_by_chunk(oi.pack_simple_embeddings([e[0] for e in embs]))
    np.testing.assert_array_equal(idx.chunk_ids, [0, 1, 2])
    assert idx.embeddings.shape == (3, 4) and idx.embeddings.dtype == np.float32
    pages = oi.pack_multi_embeddings([0, 2, 2], [np.full(4, 1.0), np.full(4, 2.0), np.full(4, 3.0)], 3)
    assert [len(p) for p in pages] == [1, 0, 2]
    byp = oi.create_index_by_page([1, 3, 3, 2], pages)
    np.testing.assert_array_equal(byp.chunk_ids, [0, 1, 1, 2, 2])
    assert oi.create_index_by_chunk(None).embeddings.size == 0
    assert oi.create_index_by_page([1], None).embeddings.size == 0
