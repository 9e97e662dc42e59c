"""GPU parity: the HIP vector path (through the C ABI) against the CPU oracle,
the golden vectors recorded from the reference, and the reference's own
known-answer / tie-break cases.  Tolerances: ids identical (bit-exact index
work); distances within 1e-4 as north_star states (observed ~1e-12 on the
float64 path, asserted tighter where the arithmetic allows)."""

import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

METRICS = ["cosine_sim", "euclidean_dist", "sqeuclidean_dist", "inner_product"]


@pytest.fixture(scope="module")
def amd():
    from aidial_rag_amd import _native
    from aidial_rag_amd.index_record import RetrievalType, to_metadata_doc
    from aidial_rag_amd.retrievers import embeddings_index as ei
    from aidial_rag_amd.retrievers import embeddings_metrics as em

    assert _native.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"

    class NS:
        pass

    ns = NS()
    ns.nat, ns.ei, ns.em, ns.RetrievalType, ns.to_metadata_doc = _native, ei, em, RetrievalType, to_metadata_doc
    return ns


def unit(x):
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)


# cosine only: the reference normalises float32 doc rows in float32 through torch, whose norm
# reduction order is CPU-specific; its own scores carry ~1e-7 noise (see test_oracle_metrics),
# so ids are compared up to ties within that noise.  Every other metric: ids identical.
COS_NOISE = 2e-7


def assert_same_ids(metric, got_rows, want_rows, oracle_dist_of_row, msg=""):
    got_rows, want_rows = np.asarray(got_rows), np.asarray(want_rows)
    assert len(got_rows) == len(want_rows), msg
    if metric != "cosine_sim":
        np.testing.assert_array_equal(got_rows, want_rows, err_msg=msg)
        return
    for g, w in zip(got_rows, want_rows):
        if g != w:
            assert abs(oracle_dist_of_row(g) - oracle_dist_of_row(w)) <= COS_NOISE, f"{msg}: {g} vs {w}"


# ---------------------------------------------------------------- metrics

def test_metric_known_answers(amd, golden_dir):
    data = json.load(open(os.path.join(golden_dir, "metrics_known.json")))
    for c in data["cases"]:
        q = np.array(c["query"], dtype=c["query_dtype"])
        d = np.array(c["docs"], dtype=c["docs_dtype"])
        out = amd.em.ENUM_TO_METRIC[amd.em.Metric(c["metric"])](q, d)
        np.testing.assert_allclose(out, np.array(c["asserted"]), rtol=1e-7, atol=0)
        assert str(out.dtype) == c["reference_dtype"]
    with pytest.raises(ValueError):
        amd.em.Metric("bad")


@pytest.mark.parametrize("tag", ["384", "1024"])
@pytest.mark.parametrize("metric", METRICS)
def test_metric_random_vs_reference_golden(amd, golden_dir, tag, metric):
    z = np.load(os.path.join(golden_dir, "metrics_random.npz"))
    docs, qs = z[f"docs{tag}"], z[f"q{tag}"]
    f = amd.em.ENUM_TO_METRIC[amd.em.Metric(metric)]
    scale = 1.0 if tag == "384" else 1024.0
    for i, q in enumerate(qs):
        o64 = f(q.astype(np.float64), docs)
        assert o64.dtype == np.float64
        atol = 1.5e-7 if metric == "cosine_sim" else 1e-11 * scale
        np.testing.assert_allclose(o64, z[f"{metric}_{tag}_f64"][i], rtol=0, atol=atol)
        o32 = f(q, docs)
        assert o32.dtype == np.float32
        np.testing.assert_allclose(o32, z[f"{metric}_{tag}_f32"][i], rtol=0, atol=1e-4 * scale)


@pytest.mark.parametrize("metric", METRICS)
def test_metric_ties_nan_zero(amd, golden_dir, metric):
    z = np.load(os.path.join(golden_dir, "metrics_ties.npz"))
    docs, queries = z["docs"], z["queries"]
    f = amd.em.ENUM_TO_METRIC[amd.em.Metric(metric)]
    for i, q in enumerate(queries):
        o = f(q.astype(np.float64), docs)
        ref = z[f"{metric}_f64"][i]
        np.testing.assert_array_equal(np.isnan(o), np.isnan(ref))
        # euclid: sqrt amplifies the 1e-16 summation-order noise of a ~1e-8 squared distance
        atol = {"cosine_sim": 1.5e-7, "euclidean_dist": 1e-10}.get(metric, 1e-12)
        np.testing.assert_allclose(o, ref, rtol=0, atol=atol, equal_nan=True)


def test_doc_sq_is_numpy_bit_exact(amd):
    """sqeuclid with a zero query is exactly float64(np.sum(docs**2, axis=1))."""
    rng = np.random.default_rng(11)
    for d in (3, 7, 8, 13, 100, 128, 129, 384, 1000, 1024):
        docs = rng.standard_normal((77, d)).astype(np.float32)
        out = amd.em.ENUM_TO_METRIC[amd.em.Metric.SQEUCLIDEAN_DIST](np.zeros(d), docs)
        np.testing.assert_array_equal(out, np.sum(docs**2, axis=1).astype(np.float64))


# ---------------------------------------------------------------- index

def test_reference_index_cases(amd, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "index_cases.json")))
    docs = {
        k: amd.ei.DocIndex(np.array(v["chunk_ids"], dtype=np.int64), np.array(v["embeddings"], dtype=np.float32))
        for k, v in cases["docs"].items()
    }
    for case in cases["cases"]:
        for metric in cases["metrics"]:
            ix = amd.ei.EmbeddingsIndex(
                amd.RetrievalType.TEXT, [docs[d] for d in case["doc_order"]], metric=amd.em.Metric(metric), limit=case["limit"]
            )
            got = ix.find(np.array(case["query"]))
            want = [amd.to_metadata_doc(a, b, amd.RetrievalType.TEXT) for a, b in case["expected"]]
            assert got == want, (case["name"], metric, got)


def test_unknown_metric_raises(amd):
    ix = amd.ei.EmbeddingsIndex(amd.RetrievalType.TEXT, [amd.ei.DocIndex(np.array([0]), np.ones((1, 3), np.float32))], metric="bad", limit=1)
    with pytest.raises(ValueError):
        ix.find(np.ones(3))


@pytest.mark.parametrize("metric", METRICS)
def test_ties_resolve_like_stable_argsort(amd, golden_dir, metric):
    from oracle import embeddings_index as oi

    z = np.load(os.path.join(golden_dir, "metrics_ties.npz"))
    docs, queries = z["docs"], z["queries"]
    dev = amd.ei.DeviceIndex.from_host(docs)
    for k in (1, 3, 10, 40, 56):
        _, _, rows, dist, cnt, _ = dev.search(queries.astype(np.float64), k, metric)
        for i, q in enumerate(queries):
            with np.errstate(invalid="ignore"):
                wrows, wdist = oi.find_flat(q.astype(np.float64), docs, metric, k)
            n_ok = int(cnt[i])
            assert n_ok == len(wrows)
            if metric == "cosine_sim" and i == 3:
                continue  # zero query: every cosine is +-0, order is by row; checked below
            with np.errstate(invalid="ignore"):
                alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](q.astype(np.float64), docs)
            assert_same_ids(metric, rows[i, :n_ok], wrows, lambda r: alld[r], f"{metric} k={k} q={i}")
            np.testing.assert_allclose(dist[i, :n_ok], wdist, rtol=0, atol=2e-7, equal_nan=True)
    # all-equal distances (zero query under cosine) come back in row order
    _, _, rows, dist, cnt, _ = dev.search(np.zeros((1, 384)), 10, "cosine_sim")
    np.testing.assert_array_equal(rows[0], np.arange(10))
    assert (dist[0] == 0).all()


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,d", [(1, 3), (5, 3), (31, 16), (33, 17), (1000, 100), (4097, 384), (700, 1024), (300, 520)])
def test_shapes_vs_oracle(amd, metric, n, d):
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(n * 1000 + d)
    docs = rng.standard_normal((n, d)).astype(np.float32)
    if metric in ("cosine_sim", "inner_product") or n % 2:
        docs = unit(docs)
    qs = rng.standard_normal((5, d))
    dev = amd.ei.DeviceIndex.from_host(docs)
    for k in (1, 7, 10):
        _, chunk, rows, dist, cnt, flags = dev.search(qs, k, metric)
        for i, q in enumerate(qs):
            wrows, wdist = oi.find_flat(q, docs, metric, k)
            assert cnt[i] == len(wrows) == min(k, n)
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](q, docs)
            assert_same_ids(metric, rows[i, : cnt[i]], wrows, lambda r: alld[r], f"{metric} n={n} d={d} k={k}")
            np.testing.assert_array_equal(chunk[i, : cnt[i]], rows[i, : cnt[i]])
            np.testing.assert_allclose(dist[i, : cnt[i]], wdist, rtol=1e-9, atol=1e-4)
            np.testing.assert_allclose(dist[i, : cnt[i]], wdist, rtol=1e-6, atol=2e-7 * d)


@pytest.mark.parametrize("metric", METRICS)
def test_many_documents_pairs_vs_oracle(amd, metric):
    """(doc_id, chunk_id) pairs over 300 ragged documents incl. empty ones and a page-style index with repeated chunk ids."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(42)
    sizes = rng.integers(0, 40, 300)
    parts = [unit(rng.standard_normal((m, 384))) if m else np.zeros((0, 384), np.float32) for m in sizes]
    parts[7][3] = parts[2][0]
    parts[200][1] = parts[2][0]
    chunk_ids = [np.sort(rng.integers(0, max(1, m // 2 + 1), m)).astype(np.int64) for m in sizes]  # repeats
    mine = [amd.ei.DocIndex(c, p) if len(p) else amd.ei.DocIndex() for c, p in zip(chunk_ids, parts)]
    theirs = [oi.DocIndex(c, p) if len(p) else oi.DocIndex() for c, p in zip(chunk_ids, parts)]
    qs = np.concatenate([parts[2][0][None].astype(np.float64), rng.standard_normal((7, 384))])
    ix = amd.ei.EmbeddingsIndex(amd.RetrievalType.IMAGE, mine, metric=metric, limit=7)
    got = ix.find_batch(qs)
    single = [ix.find(q) for q in qs]
    assert got == single  # B=1 and B=8 passes agree
    for q, res in zip(qs, got):
        with np.errstate(invalid="ignore"):
            want, _ = oi.find(q, theirs, metric, 7)
        assert [(d.metadata["doc_id"], d.metadata["chunk_id"]) for d in res] == want
        assert all(d.metadata["retrieval_type"] == amd.RetrievalType.IMAGE for d in res)
        assert all(d.page_content == f'{d.metadata["doc_id"]}_{d.metadata["chunk_id"]}' for d in res)


def test_batches_larger_than_a_query_group(amd):
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(3)
    docs = unit(rng.standard_normal((20000, 384)))
    qs = unit(rng.standard_normal((70, 384))).astype(np.float64)
    dev = amd.ei.DeviceIndex.from_host(docs)
    _, _, rows, dist, cnt, flags = dev.search(qs, 10, "sqeuclidean_dist")
    assert (cnt == 10).all() and (flags == 0).all()
    for i in range(70):
        wrows, wdist = oi.find_flat(qs[i], docs, "sqeuclidean_dist", 10)
        np.testing.assert_array_equal(rows[i], wrows)
        np.testing.assert_allclose(dist[i], wdist, rtol=0, atol=1e-12)


@pytest.mark.parametrize("metric", ["sqeuclidean_dist", "cosine_sim", "inner_product"])
def test_large_shard_wide_scan_with_sample_thresholds(amd, metric):
    """1.2M rows x 128 queries: the 128-query scan (16 queries per wave) with its sample pre-pass and the two-launch
    progressive thresholds (shards of >= 64 tiles per workgroup).  Oracle on 6 queries (the CPU path costs ~1 s per query here); planted
    duplicates must resolve to the lower row; every flag must be clear; repeated runs must agree bit for bit
    (the scan has inter-wave hand-offs: a race would show up as run-to-run differences)."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(77)
    n = 1_200_000
    docs = rng.standard_normal((n, 384), dtype=np.float32)
    docs /= np.linalg.norm(docs, axis=1, keepdims=True)
    qs = unit(rng.standard_normal((128, 384))).astype(np.float64)
    docs[900_001] = docs[77]
    docs[1_199_999] = docs[77]
    qs[0] = docs[77]
    dev = amd.ei.DeviceIndex.from_host(docs)
    runs = [dev.search(qs, 10, metric) for _ in range(3)]
    for r in runs[1:]:
        np.testing.assert_array_equal(r[2], runs[0][2])
        np.testing.assert_array_equal(r[3], runs[0][3])
    _, _, rows, dist, cnt, flags = runs[0]
    assert (cnt == 10).all()
    assert flags[1:].sum() == 0  # query 0 ties exactly at the cut by construction
    assert list(rows[0, :3]) == [77, 900_001, 1_199_999]
    for i in range(6):
        wrows, wdist = oi.find_flat(qs[i], docs, metric, 10)
        alld = None
        if metric == "cosine_sim":
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs)
        assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} q={i}")
        np.testing.assert_allclose(dist[i], wdist, rtol=0, atol=2e-7)
    # a smaller batch of the same queries rides the same 16-queries-per-wave kernels (fewer waves active): same answers
    _, _, rows32, dist32, _, _ = dev.search(qs[:32], 10, metric)
    np.testing.assert_array_equal(rows32, rows[:32])
    np.testing.assert_array_equal(dist32, dist[:32])


@pytest.mark.parametrize("metric", ["sqeuclidean_dist", "cosine_sim"])
def test_float16_storage_d1024_multimodal_shape(amd, metric):
    """BASELINE config 5 shape: d = 1024 float16 vectors, NOT normalised.  The oracle sees the float16 values
    widened to float32; the GPU index is built straight from the float16 matrix."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(2024)
    docs16 = rng.standard_normal((30000, 1024)).astype(np.float16)
    docs16[12345] = docs16[77]
    qs = rng.standard_normal((40, 1024))
    qs[0] = docs16[77].astype(np.float64)
    dev = amd.ei.DeviceIndex.from_host(docs16)
    _, _, rows, dist, cnt, flags = dev.search(qs, 10, metric)
    docs32 = docs16.astype(np.float32)
    assert list(rows[0, :2]) == [77, 12345]
    for i in range(40):
        wrows, wdist = oi.find_flat(qs[i], docs32, metric, 10)
        alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs32) if metric == "cosine_sim" else None
        assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} q={i}")
        np.testing.assert_allclose(dist[i], wdist, rtol=1e-12, atol=2e-7)
    if metric != "cosine_sim":  # full-vector API on the float16-built index; differs only by float64 summation order
        np.testing.assert_allclose(dev.metric_eval(qs[1], metric), oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[1], docs32), rtol=1e-13)


@pytest.mark.parametrize("metric", ["inner_product", "euclidean_dist", "sqeuclidean_dist", "cosine_sim"])
def test_float16_native_scan_shapes(amd, metric):
    """The float16-native path (d = 1024 kept as 2-byte fragments, 128 queries per pass, 16 per wave, one float16 product
    per fragment - vec_kernels_h16.h): a ragged last tile, batches of 1 / 33 / 64 / 100 / 130 queries (one and two passes,
    idle and partly filled waves), duplicates across the tile boundary, queries of very different magnitude (per-query
    power-of-two scaling)."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(515)
    n = 20011
    docs16 = (rng.standard_normal((n, 1024)) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float16)
    docs16[31] = docs16[32] = docs16[n - 1]
    docs32 = docs16.astype(np.float32)
    dev = amd.ei.DeviceIndex.from_host(docs16)
    assert dev.hbm_bytes() < 2.2 * docs16.nbytes  # rows (2 B) + fragments (2 B) + norms: not the 8 x of the widened path
    qs = rng.standard_normal((130, 1024))
    qs[3] *= 1e-6
    qs[4] *= 3e4
    qs[5] = docs32[n - 1].astype(np.float64)
    for b in (1, 33, 64, 100, 130):
        _, _, rows, dist, cnt, flags = dev.search(qs[:b], 10, metric)
        assert (cnt == 10).all()
        for i in range(b):
            wrows, wdist = oi.find_flat(qs[i], docs32, metric, 10)
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs32) if metric == "cosine_sim" else None
            assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} b={b} q={i}")
            np.testing.assert_allclose(dist[i], wdist, rtol=1e-12, atol=2e-7 * max(1.0, float(np.abs(wdist).max())))
    if metric == "sqeuclidean_dist":
        assert list(rows[5, :3]) == [31, 32, n - 1]
    # k up to the float16 scan's list capacity (48: lists of k + 12 <= 60) runs the filter scan - whose bound
    # (2^-11 |q| max|d|, one float16 product per fragment) may or may not prove a query on rows whose norms differ 15-fold
    # as here; above it the exact pass answers alone.  Either way the ids are the reference's.
    for k, must_be_exact in ((48, False), (49, True)):
        _, _, rows_k, _, cnt_k, flags_k = dev.search(qs[:2], k, metric)
        assert (cnt_k == k).all() and flags_k[1] in ((amd.nat.FLAG_EXACT_PASS,) if must_be_exact else (0, amd.nat.FLAG_EXACT_PASS))
        for i in range(2):
            wrows, _ = oi.find_flat(qs[i], docs32, metric, k)
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs32) if metric == "cosine_sim" else None
            assert_same_ids(metric, rows_k[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} k={k} q={i}")


@pytest.mark.parametrize("d", [257, 384, 512, 520])
def test_float16_native_narrower_dims(amd, d):
    """Float16 rows with 256 < d <= 1024 stay 2 bytes per element: d <= 512 is one ring stage per tile (16 k-steps of 32
    columns, zero-padded), d = 520 the two-stage form with a mostly empty second half; odd d takes the element-wise
    packing path."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(100 + d)
    n = 9000
    docs16 = rng.standard_normal((n, d)).astype(np.float16)
    docs16[4000] = docs16[17]
    docs32 = docs16.astype(np.float32)
    dev = amd.ei.DeviceIndex.from_host(docs16)
    assert dev.hbm_bytes() < (2 * 2 * (512 if d <= 512 else 1024) + 64) * n  # rows + padded fragments + norms
    qs = rng.standard_normal((40, d))
    qs[7] = docs32[17].astype(np.float64)
    for metric in ("sqeuclidean_dist", "cosine_sim", "inner_product"):
        _, _, rows, dist, cnt, flags = dev.search(qs, 10, metric)
        assert (cnt == 10).all()
        for i in range(40):
            wrows, wdist = oi.find_flat(qs[i], docs32, metric, 10)
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs32) if metric == "cosine_sim" else None
            assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} d={d} q={i}")
            np.testing.assert_allclose(dist[i], wdist, rtol=1e-12, atol=2e-7 * max(1.0, float(np.abs(wdist).max())))
        if metric == "sqeuclidean_dist":
            assert list(rows[7, :2]) == [17, 4000]


@pytest.mark.parametrize("metric", METRICS)
def test_float16_native_vs_reference_golden(amd, golden_dir, metric):
    """The d = 1024 fp16-rounded set of metrics_random.npz (outputs recorded from the imported reference
    module): built as a float16-NATIVE index, full-vector metric and top-k against the reference's values."""
    z = np.load(os.path.join(golden_dir, "metrics_random.npz"))
    docs32, qs = z["docs1024"], z["q1024"].astype(np.float64)
    docs16 = docs32.astype(np.float16)
    assert np.array_equal(docs16.astype(np.float32), docs32)  # the fixture IS fp16-representable
    dev = amd.ei.DeviceIndex.from_host(docs16)
    _, _, rows, dist, cnt, _ = dev.search(qs, 10, metric)
    for i in range(len(qs)):
        ref = z[f"{metric}_1024_f64"][i]
        got = dev.metric_eval(qs[i], metric)
        atol = 1.5e-7 if metric == "cosine_sim" else 1e-8
        np.testing.assert_allclose(got, ref, rtol=0, atol=atol)
        want = np.argsort(ref, kind="stable")[:10]
        assert_same_ids(metric, rows[i], want, lambda r: ref[r], f"{metric} q={i}")
        np.testing.assert_allclose(dist[i], ref[want], rtol=0, atol=atol)


def test_float16_native_padded_dimension(amd):
    """d = 840 is not a multiple of 512: the fragment copy is zero-padded to 1024 columns, norms and the
    re-score use the true d."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(1408)
    docs16 = rng.standard_normal((9000, 840)).astype(np.float16)
    docs32 = docs16.astype(np.float32)
    qs = rng.standard_normal((70, 840))
    dev = amd.ei.DeviceIndex.from_host(docs16)
    assert dev.hbm_bytes() < 2.4 * docs16.nbytes  # rows + fragments padded 840 -> 1024
    for metric in ("sqeuclidean_dist", "cosine_sim", "inner_product"):
        _, _, rows, dist, cnt, flags = dev.search(qs, 10, metric)
        for i in range(0, 70, 7):
            wrows, wdist = oi.find_flat(qs[i], docs32, metric, 10)
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs32) if metric == "cosine_sim" else None
            assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} q={i}")
            np.testing.assert_allclose(dist[i], wdist, rtol=1e-12, atol=1e-6)


def test_float16_native_large_shard_sample_prepass(amd):
    """More than 32768 tiles: the threshold pre-pass runs on the float16 scan too."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(616)
    n = 1_060_000
    docs16 = np.empty((n, 1024), np.float16)
    for c in range(0, n, 106_000):
        docs16[c : c + 106_000] = rng.standard_normal((106_000, 1024), dtype=np.float32).astype(np.float16)
    dev = amd.ei.DeviceIndex.from_host(docs16)
    qs = rng.standard_normal((70, 1024))
    _, _, rows, dist, cnt, flags = dev.search(qs, 10, "sqeuclidean_dist")
    assert (flags == 0).all()
    fn = oi.ENUM_TO_METRIC[oi.Metric("sqeuclidean_dist")]
    for i in (0, 33, 69):
        alld = np.concatenate([fn(qs[i], docs16[c : c + 106_000].astype(np.float32)) for c in range(0, n, 106_000)])
        want = np.argsort(alld, kind="stable")[:10]
        np.testing.assert_array_equal(rows[i], want)
        np.testing.assert_allclose(dist[i], alld[want], rtol=1e-12, atol=1e-6)


@pytest.mark.parametrize("d", [384, 48])
@pytest.mark.parametrize("metric", ["sqeuclidean_dist", "cosine_sim"])
def test_register_ring_kernel_large_k(amd, metric, d):
    """k = 30: at d = 384 the 128-query kernel with long candidate buffers (klist 38: four ring stages instead of five), at
    d = 48 the 32-query register-ring kernel (scan_topk_kernel) with 38-entry per-lane lists."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(77)
    docs = unit(rng.standard_normal((40000, d)))
    qs = unit(rng.standard_normal((40, d))).astype(np.float64)
    dev = amd.ei.DeviceIndex.from_host(docs)
    _, _, rows, dist, cnt, flags = dev.search(qs, 30, metric)
    assert (cnt == 30).all()
    for i in range(40):
        wrows, wdist = oi.find_flat(qs[i], docs, metric, 30)
        alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs) if metric == "cosine_sim" else None
        assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} q={i}")
        np.testing.assert_allclose(dist[i], wdist, rtol=0, atol=2e-7)


def test_k_above_list_capacity_takes_the_exact_pass(amd):
    """k beyond the filter's candidate lists is answered by the exact pass alone (tests/test_gpu_exact_pass.py)."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(4)
    docs = unit(rng.standard_normal((500, 32)))
    dev = amd.ei.DeviceIndex.from_host(docs)
    q = rng.standard_normal((1, 32))
    _, _, rows, _, cnt, flags = dev.search(q, 100, "inner_product")
    assert cnt[0] == 100 and flags[0] == amd.nat.FLAG_EXACT_PASS
    np.testing.assert_array_equal(rows[0], oi.find_flat(q[0], docs, "inner_product", 100)[0])
    small = amd.ei.DeviceIndex.from_host(unit(rng.standard_normal((40, 32))))
    _, _, rows, _, cnt, _ = small.search(rng.standard_normal((1, 32)), 100, "inner_product")
    assert cnt[0] == 40 and sorted(rows[0, :40]) == list(range(40))


def test_concurrent_searches_share_a_handle(amd):
    import threading

    rng = np.random.default_rng(8)
    docs = unit(rng.standard_normal((30000, 384)))
    dev = amd.ei.DeviceIndex.from_host(docs)
    qs = rng.standard_normal((16, 384))
    want = dev.search(qs, 7, "cosine_sim")[2]
    out = [None] * 8

    def work(t):
        out[t] = [dev.search(qs[2 * t : 2 * t + 2], 7, "cosine_sim")[2] for _ in range(5)]

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for t in range(8):
        for r in out[t]:
            np.testing.assert_array_equal(r, want[2 * t : 2 * t + 2])


def test_concurrent_find_calls_share_passes(amd):
    """EmbeddingsIndex.find from many threads (the reference's executor threads, semantic_retriever.py:54-56):
    answers equal the sequential ones, and the calls were served by fewer search passes than calls."""
    import threading

    rng = np.random.default_rng(12)
    docs = unit(rng.standard_normal((200_000, 384)))
    index = amd.ei.EmbeddingsIndex(
        retrieval_type="text",
        indexes=[amd.ei.DocIndex(chunk_ids=np.arange(100_000), embeddings=docs[:100_000]),
                 amd.ei.DocIndex(chunk_ids=np.arange(100_000), embeddings=docs[100_000:])],
        limit=7,
    )
    qs = unit(rng.standard_normal((64, 384))).astype(np.float64)
    key = lambda found: [(d.metadata["doc_id"], d.metadata["chunk_id"]) for d in found]
    want = [key(index.find(q)) for q in qs]
    passes0, calls0 = index._commit.passes, index._commit.calls
    got = [None] * 64

    def work(t):
        for i in range(t, 64, 16):
            got[i] = key(index.find(qs[i]))

    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == want
    assert index._commit.calls - calls0 == 64
    assert index._commit.passes - passes0 < 64


def test_merge_device_matches_host(amd):
    import ctypes as C

    import torch

    rng = np.random.default_rng(9)
    s, b, k = 4, 6, 5
    dist = np.round(rng.standard_normal((s, b, k)), 1)
    dist[1, 2, 3] = np.nan
    dist.sort(axis=2)
    row = rng.permutation(s * b * k).reshape(s, b, k).astype(np.int64)
    cnt = rng.integers(0, k + 1, (s, b)).astype(np.int32)
    for desc in (0, 1):
        hd, hr, hc = np.zeros((b, k)), np.zeros((b, k), np.int64), np.zeros(b, np.int32)
        amd.nat.check(amd.nat.lib.mir_topk_merge_host(amd.nat.ptr(dist), amd.nat.ptr(row), amd.nat.ptr(cnt), s, 0, b, k, desc,
                                                      amd.nat.ptr(hd), amd.nat.ptr(hr), amd.nat.ptr(hc)))
        td, tr, tc = (torch.from_numpy(x).cuda() for x in (dist, row, cnt))
        od = torch.zeros((b, k), dtype=torch.float64, device="cuda")
        orow = torch.zeros((b, k), dtype=torch.int64, device="cuda")
        oc = torch.zeros(b, dtype=torch.int32, device="cuda")
        amd.nat.check(amd.nat.lib.mir_topk_merge_device(td.data_ptr(), tr.data_ptr(), tc.data_ptr(), s, 0, b, k, desc,
                                                        od.data_ptr(), orow.data_ptr(), oc.data_ptr(), 0,
                                                        torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(oc.cpu().numpy(), hc)
        for q in range(b):
            np.testing.assert_array_equal(orow.cpu().numpy()[q, : hc[q]], hr[q, : hc[q]])
            np.testing.assert_array_equal(od.cpu().numpy()[q, : hc[q]], hd[q, : hc[q]])


# ---------------------------------------------------------------- full size (BASELINE config C4 on one GPU)

def oracle_topk_chunked(emb, q, metric, k, chunk=500_000):
    """The float64 oracle over a matrix that lives on the GPU, in row chunks on the host: per-chunk stable top-k (global
    row ids), chunks concatenated in order, one more stable argsort - the reference's own two-level structure
    (embeddings_index.py:62-89 upstream) with chunks as documents, hence identical to one global stable sort.
    Returns (rows[k], dist[k], dist_of_row callable)."""
    from oracle import embeddings_index as oi

    f = oi.ENUM_TO_METRIC[oi.Metric(metric)]
    cand_rows, cand_dist = [], []
    n = emb.shape[0]
    for c in range(0, n, chunk):
        blk = emb[c : c + chunk].float().cpu().numpy()
        with np.errstate(invalid="ignore"):
            dist = f(q, blk)
        top = np.argsort(dist, kind="stable")[:k]
        cand_rows.append(top + c)
        cand_dist.append(dist[top])
    cr, cd = np.concatenate(cand_rows), np.concatenate(cand_dist)
    top = np.argsort(cd, kind="stable")[:k]

    def dist_of_row(r):
        return float(f(q, emb[int(r) : int(r) + 1].float().cpu().numpy())[0])

    return cr[top].astype(np.int64), cd[top], dist_of_row


def test_full_size_10m_properties(amd):
    """10M x 384 float32 - the size the headline metric is quoted on - checked through properties that
    do not need the CPU oracle to scan 15 GB for every query: 256 queries = ONE launch of the headline shape (two 16-query
    tiles per wave, QT = 2); planted copies of the queries are found first at distance 0; returned distances are ascending
    and equal the float64 formula on the returned rows; for EIGHT queries - from both query tiles of a wave - ids, order and
    distances equal the float64 oracle over the whole matrix (row chunks on the host); and the result is identical to
    searching two 5M-row shards and merging them with the library's merge."""
    torch = pytest.importorskip("torch")
    n, d, k = 10_000_000, 384, 10
    dev0 = torch.device("cuda", 0)
    emb = torch.empty((n, d), dtype=torch.float32, device=dev0)
    g = torch.Generator(device=dev0)
    for c in range(0, n, 500_000):
        g.manual_seed(99 + c)
        x = torch.randn((500_000, d), generator=g, dtype=torch.float32, device=dev0)
        emb[c : c + 500_000] = x / x.norm(dim=1, keepdim=True)
    rng = np.random.default_rng(5)
    nq = 256
    q32 = unit(rng.standard_normal((nq, d))).astype(np.float32)
    planted = [123, 4_999_999, 5_000_000, 9_999_999]
    planted_q = [0, 1, 130, 255]  # queries of both tiles of a wave
    for j, r in zip(planted_q, planted):
        emb[r] = torch.from_numpy(q32[j]).to(dev0)
    torch.cuda.synchronize()
    qs = q32.astype(np.float64)
    full = amd.ei.DeviceIndex.from_device_ptr(emb.data_ptr(), n, d, 0)
    _, _, rows, dist, cnt, flags = full.search(qs, k, "sqeuclidean_dist")
    assert (cnt == k).all() and (flags == 0).all()
    assert (np.diff(dist, axis=1) >= 0).all()
    for j, r in zip(planted_q, planted):
        assert rows[j, 0] == r and abs(dist[j, 0]) < 1e-6
    # float64 formula on the returned rows (doc_sq in float32, numpy order, as the reference computes it)
    for i in (0, 17, 99, 128, 200, 255):
        got = emb[torch.from_numpy(rows[i]).to(dev0)].cpu().numpy()
        want = np.sum(got**2, axis=1).astype(np.float64) - 2.0 * (got.astype(np.float64) @ qs[i]) + float(qs[i] @ qs[i])
        np.testing.assert_allclose(dist[i], want, rtol=0, atol=1e-12)
    # ids AND order against the float64 oracle over all 10M rows (computed in row chunks on the host, ~6 s per query)
    for i in (5, 98, 127, 128, 171, 200, 222, 254):
        wrows, wdist, _ = oracle_topk_chunked(emb, qs[i], "sqeuclidean_dist", k)
        np.testing.assert_array_equal(rows[i], wrows)
        np.testing.assert_allclose(dist[i], wdist, rtol=0, atol=1e-12)
    # two shards + the library merge == the full index
    half = n // 2
    lo = amd.ei.DeviceIndex.from_device_ptr(emb.data_ptr(), half, d, 0, row_offset=0)
    hi = amd.ei.DeviceIndex.from_device_ptr(emb.data_ptr() + half * d * 4, n - half, d, 0, row_offset=half)
    parts = [ix.search(qs, k, "sqeuclidean_dist") for ix in (lo, hi)]
    pd = np.ascontiguousarray(np.stack([p[3] for p in parts]))
    pr = np.ascontiguousarray(np.stack([p[2] for p in parts]))
    pc = np.ascontiguousarray(np.stack([p[4] for p in parts]))
    od, orow, oc = np.zeros((nq, k)), np.zeros((nq, k), np.int64), np.zeros(nq, np.int32)
    amd.nat.check(amd.nat.lib.mir_topk_merge_host(amd.nat.ptr(pd), amd.nat.ptr(pr), amd.nat.ptr(pc), 2, 0, nq, k, 0,
                                                  amd.nat.ptr(od), amd.nat.ptr(orow), amd.nat.ptr(oc)))
    np.testing.assert_array_equal(orow, rows)
    np.testing.assert_array_equal(od, dist)


def test_full_size_c5_float16_properties(amd):
    """One GPU's share of BASELINE config C5: 6.25M x 1024 float16, not normalised, cosine.  Same properties
    as the float32 full-size test; the float32 torch reference works on the widened float16 values."""
    torch = pytest.importorskip("torch")
    n, d, k = 6_250_000, 1024, 10
    dev0 = torch.device("cuda", 0)
    emb = torch.empty((n, d), dtype=torch.float16, device=dev0)
    g = torch.Generator(device=dev0)
    for c in range(0, n, 250_000):
        g.manual_seed(2024 + c)
        emb[c : c + 250_000] = torch.randn((250_000, d), generator=g, dtype=torch.float32, device=dev0).half()
    rng = np.random.default_rng(6)
    qs = rng.standard_normal((70, d))
    planted = [7, 3_124_999, 3_125_000, 6_249_999]
    for j, r in enumerate(planted):
        emb[r] = torch.from_numpy(qs[j].astype(np.float16)).to(dev0)
        qs[j] = qs[j].astype(np.float16).astype(np.float64) * (j + 1)  # cosine ignores the query's length
    torch.cuda.synchronize()
    full = amd.ei.DeviceIndex.from_device_ptr(emb.data_ptr(), n, d, 0, float16=True)
    assert full.hbm_bytes() < 2.1 * n * d * 2
    _, _, rows, dist, cnt, flags = full.search(qs, k, "cosine_sim")
    assert (cnt == k).all() and (flags == 0).all()
    assert (np.diff(dist, axis=1) >= 0).all()
    for j, r in enumerate(planted):
        assert rows[j, 0] == r and abs(dist[j, 0] + 1.0) < 1e-6
    # ids AND order against the oracle (torch cosine on the widened rows, in row chunks on the host); cosine ids are
    # compared up to ties within the reference's own float32-normalisation noise (COS_NOISE)
    for i in (10, 69):
        wrows, wdist, dist_of_row = oracle_topk_chunked(emb, qs[i], "cosine_sim", k, chunk=250_000)
        assert_same_ids("cosine_sim", rows[i], wrows, dist_of_row, f"C5 q={i}")
        np.testing.assert_allclose(dist[i], wdist, rtol=0, atol=2e-7)
    half = n // 2
    lo = amd.ei.DeviceIndex.from_device_ptr(emb.data_ptr(), half, d, 0, row_offset=0, float16=True)
    hi = amd.ei.DeviceIndex.from_device_ptr(emb.data_ptr() + half * d * 2, n - half, d, 0, row_offset=half, float16=True)
    parts = [ix.search(qs, k, "cosine_sim") for ix in (lo, hi)]
    pd = np.ascontiguousarray(np.stack([p[3] for p in parts]))
    pr = np.ascontiguousarray(np.stack([p[2] for p in parts]))
    pc = np.ascontiguousarray(np.stack([p[4] for p in parts]))
    od, orow, oc = np.zeros((70, k)), np.zeros((70, k), np.int64), np.zeros(70, np.int32)
    amd.nat.check(amd.nat.lib.mir_topk_merge_host(amd.nat.ptr(pd), amd.nat.ptr(pr), amd.nat.ptr(pc), 2, 0, 70, k, 0,
                                                  amd.nat.ptr(od), amd.nat.ptr(orow), amd.nat.ptr(oc)))
    np.testing.assert_array_equal(orow, rows)
    np.testing.assert_array_equal(od, dist)


@pytest.mark.parametrize("n", [40_003, 20_011])
@pytest.mark.parametrize("metric,d", [(m, 1024) for m in METRICS] + [("sqeuclidean_dist", 768), ("cosine_sim", 520), ("inner_product", 400)])
def test_float32_wide_dims_k_split_scan(amd, metric, d, n):
    """float32 rows with 384 < d <= 1024 (multimodal / description page embeddings are stored as float32,
    embeddings_index.py:139-153 upstream), a ragged last tile, duplicates, batches of 1 / 64 / 100.  40 003 rows (>= 32K): the
    sieve's bf16 filter over the hi-only image (round 4: sieve_h16_kernel<BF>, d padded to 512 / 1024) - every query answered
    with flag 0; 20 011 rows: the 64-query K-split list scan over the bf16 hi/lo image (d padded to 512 / 768 / 1024)."""
    from oracle import embeddings_index as oi

    rng = np.random.default_rng(1000 + d)
    docs = (rng.standard_normal((n, d)) * rng.uniform(0.3, 2.0, (n, 1))).astype(np.float32)
    if metric in ("cosine_sim", "inner_product"):
        docs = unit(docs)
    docs[31] = docs[32] = docs[n - 1]
    qs = rng.standard_normal((100, d))
    qs[5] = docs[n - 1].astype(np.float64)
    dev = amd.ei.DeviceIndex.from_host(docs)
    dev.scan_stats(reset=True)
    for b in (1, 64, 100):
        _, _, rows, dist, cnt, flags = dev.search(qs[:b], 10, metric)
        assert (cnt == 10).all()
        if n >= 32768:
            assert (flags == 0).all(), f"{metric} d={d}: the wide sieve needed the exact pass: {flags}"
        for i in ([0, 5] if b == 1 else [1, 5, 33, 63]) [: b] + ([99] if b == 100 else []):
            with np.errstate(invalid="ignore"):
                wrows, wdist = oi.find_flat(qs[i], docs, metric, 10)
                alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs) if metric == "cosine_sim" else None
            assert_same_ids(metric, rows[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} d={d} b={b} q={i}")
            np.testing.assert_allclose(dist[i], wdist, rtol=1e-9, atol=2e-7 * d, equal_nan=True)
    assert (dev.scan_stats()["queries"] > 0) == (n >= 32768)  # the sieve's counters: it ran iff the shard is large enough
    if metric in ("sqeuclidean_dist", "inner_product"):
        _, _, rows, _, _, _ = dev.search(qs[5:6], 10, metric)
        assert list(rows[0, :3]) == [31, 32, n - 1]
    # k beyond the 64-query kernel's lists (28): the 32-query kernels take over, same answers
    _, _, rows40, _, cnt40, _ = dev.search(qs[:3], 40, metric)
    for i in range(3):
        with np.errstate(invalid="ignore"):
            wrows, _ = oi.find_flat(qs[i], docs, metric, 40)
            alld = oi.ENUM_TO_METRIC[oi.Metric(metric)](qs[i], docs) if metric == "cosine_sim" else None
        assert_same_ids(metric, rows40[i], wrows, (lambda r: alld[r]) if alld is not None else None, f"{metric} d={d} k=40 q={i}")
