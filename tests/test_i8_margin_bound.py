"""The bound behind the sieve's int8 first stage (csrc/vec_kernels_i8.h), restated in numpy and checked on the CPU: with one
scale per 32-row tile and one per query, I = X.Q exact,

    |x.q - s_t s_q I|  <=  e_t (|q| + e_q) + |x|max e_q          e_t = the tile's largest |x - s_t X|,  e_q = |q - s_q Q|

(Cauchy-Schwarz on the two rounding residuals: i8_margin_ab / i8_margin_tile without their float32 slack).  It must hold for
every (row, query) pair of random data and be nearly ATTAINED by a row whose residual is aligned with the query - so no smaller
multiple of it would do (the GPU tests test_worst_case_int8_rounding and tools/worst_case_margin_check.sh show the same on the
device)."""

import numpy as np


def quantise_rows(x):
    """(X int32 [n, d], s [n]): int8 grid per 32-row tile, scale = the tile's largest |x_i| / 127, round to nearest, float32 as on the device"""
    n, d = x.shape
    tiles = x.reshape(n // 32, 32, d)
    m = np.abs(tiles).max(axis=(1, 2)).astype(np.float32)
    s = np.where(m > 0, m / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    inv = (np.float32(1.0) / s).astype(np.float32)
    X = np.clip(np.rint(tiles * inv[:, None, None]), -127, 127).astype(np.int32).reshape(n, d)
    return X, np.repeat(s, 32)


def quantise_queries(q):
    m = np.abs(q).max(axis=1).astype(np.float32)
    inv = np.where(m > 0, np.float32(127.0) / m, np.float32(1.0)).astype(np.float32)
    s = (np.float32(1.0) / inv).astype(np.float32)
    Q = np.clip(np.rint(q.astype(np.float32) * inv[:, None]), -127, 127).astype(np.int32)
    return Q, s


def bounds(x, q):
    X, s = quantise_rows(x)
    Q, sq = quantise_queries(q)
    xd, qd = x.astype(np.float64), q.astype(np.float64)
    e_row = np.linalg.norm(xd - s[:, None].astype(np.float64) * X, axis=1)
    e_t = np.repeat(e_row.reshape(-1, 32).max(axis=1), 32)
    e_q = np.linalg.norm(qd - sq[:, None].astype(np.float64) * Q, axis=1)
    approx = (s[:, None].astype(np.float64) * sq[None, :].astype(np.float64)) * (X.astype(np.int64) @ Q.astype(np.int64).T)
    err = np.abs(xd @ qd.T - approx)
    mg = e_t[:, None] * (np.linalg.norm(qd, axis=1) + e_q)[None, :] + np.linalg.norm(xd, axis=1).max() * e_q[None, :]
    return err, mg


def test_bound_holds_on_random_rows_and_queries():
    rng = np.random.default_rng(3)
    for d, scale in ((384, 1.0), (128, 7.5), (200, 1e-3)):
        x = rng.standard_normal((4096, d)).astype(np.float32)
        x *= np.float32(scale) / np.linalg.norm(x, axis=1, keepdims=True)
        q = rng.standard_normal((64, d)) * rng.uniform(1e-3, 40.0, (64, 1))
        err, mg = bounds(x, q)
        assert (err <= mg * (1 + 1e-9)).all(), float((err / mg).max())
        assert (err / mg).max() < 0.5  # random residuals are far from aligned: the bound is never close on such data


def test_bound_is_attained_by_an_aligned_residual():
    rng = np.random.default_rng(4)
    d = 384
    x = rng.standard_normal((64, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    sig = np.where(rng.random(d) < 0.5, -1.0, 1.0)
    q = (sig / np.sqrt(d))[None, :]                      # its int8 image is exact: e_q ~ 0
    m = np.abs(np.delete(x[:32], 7, axis=0)).max()
    s = np.float32(m) / np.float32(127.0)
    y = np.rint(rng.normal(10.0, 30.0, d))
    x[7] = (s * (y * sig + 0.49 * sig).astype(np.float32)).astype(np.float32)   # 0.49 of a step along the query on every coordinate
    assert np.abs(x[7]).max() < m
    err, mg = bounds(x, q)
    assert err[7, 0] <= mg[7, 0] * (1 + 1e-9)
    assert err[7, 0] > 0.97 * mg[7, 0]                   # Cauchy-Schwarz is tight here: half the margin would lose this row
