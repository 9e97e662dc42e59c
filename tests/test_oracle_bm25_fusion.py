"""Oracle self-consistency for the unpinned third-party pieces: BM25Okapi
(rank-bm25 0.2.2 restatement) and weighted RRF (langchain 0.3.21 restatement).
Hand-computable corpora; expected numbers worked out from the published formulas."""

import math

import numpy as np
import pytest

from oracle import bm25 as ob
from oracle import fusion as of


CORPUS = [
    ["hello", "there", "good", "man"],
    ["it", "is", "quite", "windy", "in", "london"],
    ["how", "is", "the", "weather", "today"],
    [],
    ["is", "is", "is", "london"],
]


def test_bm25_hand_computed():
    m = ob.build(CORPUS)
    assert m.corpus_size == 5 and m.avgdl == 19 / 5
    # df(is)=3 of 5 -> ln(2.5)-ln(3.5) < 0 -> floored to eps*average_idf
    raw = {w: math.log(5 - n + 0.5) - math.log(n + 0.5) for w, n in {"is": 3, "london": 2, "windy": 1}.items()}
    assert raw["is"] < 0
    assert m.idf["is"] == pytest.approx(0.25 * m.average_idf)
    assert m.idf["london"] == raw["london"] and m.idf["windy"] == raw["windy"]
    s = m.get_scores(["windy", "london", "nope", "london"])
    k1, b = 1.5, 0.75

    def term(idf, tf, dl):
        return idf * (tf * (k1 + 1) / (tf + k1 * (1 - b + b * dl / m.avgdl)))

    exp = np.zeros(5)
    exp[1] = term(raw["windy"], 1, 6) + 2 * term(raw["london"], 1, 6)
    exp[4] = 2 * term(raw["london"], 1, 4)
    np.testing.assert_allclose(s, exp, rtol=1e-15)
    assert s.dtype == np.float64


def test_bm25_topn_tie_break_goes_to_highest_index():
    m = ob.build(CORPUS)
    s = m.get_scores(["absent"])
    assert (s == 0).all()
    np.testing.assert_array_equal(ob.top_n_indexes(s, 3), [4, 3, 2])
    s = m.get_scores(["london"])
    top = ob.top_n_indexes(s, 5)
    assert set(top[:2]) == {1, 4} and list(top[2:]) == [3, 2, 0]


def test_bm25_empty_guard():
    with pytest.raises(ValueError, match="Text index is empty."):
        ob.build([[], []])
    with pytest.raises(ValueError, match="Text index is empty."):
        ob.build([])


def test_bm25_csr_matches_dict_loop():
    rng = np.random.default_rng(777)
    vocab, n = 200, 400
    lens = np.clip(np.round(rng.normal(30, 10, n)), 0, 80).astype(np.int64)
    lens[::97] = 0
    indptr = np.concatenate(([0], np.cumsum(lens)))
    toks = np.minimum(rng.zipf(1.3, int(lens.sum())) - 1, vocab - 1).astype(np.int32)
    corpus = [toks[indptr[i] : indptr[i + 1]].tolist() for i in range(n)]
    a = ob.BM25Okapi(corpus)
    c = ob.BM25OkapiCSR(indptr, toks, vocab)
    assert a.avgdl == c.avgdl and a.average_idf == c.average_idf
    for t, v in a.idf.items():
        assert c.idf[t] == v
    for q in ([0], [1, 5, 9], [3, 3, 150], [199, 7, 250, -1], [0, 1, 2, 3, 4, 5, 6, 7]):
        np.testing.assert_array_equal(a.get_scores(q), c.get_scores(q))  # bit-exact


def test_rrf_hand_computed():
    a = [(0, 1), (0, 2), (1, 0)]
    b = [(0, 2), (2, 2), (0, 2)]  # in-list duplicate is credited twice
    fused = of.weighted_reciprocal_rank([a, b], [1.0, 1.0])
    sc = of.rrf_scores([a, b], [1.0, 1.0])
    assert sc[(0, 2)] == 1 / 62 + 1 / 61 + 1 / 63
    assert sc[(0, 1)] == 1 / 61 and sc[(2, 2)] == 1 / 62 and sc[(1, 0)] == 1 / 63
    assert fused == [(0, 2), (0, 1), (2, 2), (1, 0)]
    # equal scores keep first-seen order
    assert of.weighted_reciprocal_rank([["x"], ["y"]], [1.0, 1.0]) == ["x", "y"]
    assert of.weighted_reciprocal_rank([["x"], ["y"]], [1.0, 2.0]) == ["y", "x"]
    with pytest.raises(ValueError):
        of.weighted_reciprocal_rank([["x"]], [1.0, 1.0])
    assert of.weighted_reciprocal_rank([[], []], [1.0, 1.0]) == []
