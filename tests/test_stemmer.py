"""Native Snowball-English stemmer (csrc/stem_english.cpp through the C ABI, host code - runs without a GPU)
against NLTK's SnowballStemmer("english"): the per-token step of keywords_preprocess (keywords_search.py:13-18).
Fixture: tests/golden/snowball_english.json.gz, 44k (token, stem) pairs written by make_snowball_fixture.py."""

import gzip
import json
import os

import pytest

from aidial_rag_amd import keywords_search as ks

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "snowball_english.json.gz")


@pytest.fixture(scope="module")
def pairs():
    with gzip.open(GOLDEN) as f:
        d = json.loads(f.read().decode("utf-8"))
    assert "SnowballStemmer('english')" in d["source"]
    return d["pairs"]


def test_every_golden_token(pairs):
    words = [w for w, _ in pairs]
    got = ks.stem_tokens(words)
    bad = [(w, g, s) for (w, s), g in zip(pairs, got) if g != s]
    assert not bad, bad[:20]
    assert len(pairs) > 40000


def test_one_by_one_equals_batch(pairs):
    sample = [w for w, _ in pairs[::97]]
    assert [ks.stem_tokens([w])[0] for w in sample] == ks.stem_tokens(sample)


def test_documented_quirks():
    # NLTK's special words, its R2 = "e" bookkeeping after -ational, apostrophes, y handling, short tokens
    got = ks.stem_tokens(["skies", "dying", "news", "succeeding", "relational", "dog's", "’tis", "yellow", "saying",
                          "generously", "communities", "arsenal", "is", "a", "", "Running", "ÉCOLES"])
    assert got == ["sky", "die", "news", "succeed", "relat", "dog", "tis", "yellow", "say", "generous", "communiti", "arsenal",
                   "is", "a", "", "run", "école"]


def test_batch_edges():
    assert ks.stem_tokens([]) == []
    assert ks.stem_tokens([""]) == [""]
    assert ks.stem_tokens(["", "", "cats", ""]) == ["", "", "cat", ""]
    with pytest.raises(ValueError):
        ks.stem_tokens(["a\x00b"])
    long = "internationalization" * 40
    assert ks.stem_tokens([long])[0].startswith("internationalization")


def test_keywords_preprocess_pipeline(monkeypatch):
    """Stopwords are matched BEFORE lower-casing (keywords_search.py:16-17): "The" survives, "the" does not;
    punctuation tokens are kept."""
    monkeypatch.setattr(ks, "_front_end", lambda: (frozenset({"the", "is", "of"}), str.split))
    assert ks.keywords_preprocess("The colours of the Alps , running") == ["the", "colour", "alp", ",", "run"]


def test_without_nltk_data_the_front_end_is_the_restatement():
    """No NLTK data in this image: keywords_preprocess runs on the restated tokenizer / stopword list
    (tests/test_keywords_search.py pins those); with the data installed it would use NLTK's own."""
    try:
        import nltk  # noqa: F401
        from nltk.corpus import stopwords

        stopwords.words("english")
    except (ImportError, LookupError):
        ks._front_end.cache_clear()
        stop, tokenize = ks._front_end()
        assert stop is ks.ENGLISH_STOPWORDS and tokenize is ks._word_tokenize_restated
        assert ks.keywords_preprocess("some text, surely") == ["text", ",", "sure"]


def test_compact_term_ids_first_appearance_order():
    """mir_compact_term_ids (host utility of the BM25 build): ids from a larger vocabulary -> 0..n_used-1 in order of
    first appearance, the order rank-bm25's dicts have; remap marks unused ids with -1."""
    import ctypes as C

    import numpy as np

    from aidial_rag_amd import _native as nat

    ids = np.array([7, 3, 7, 9, 3, 0, 9, 9, 5], np.int32)
    out = np.empty_like(ids)
    remap = np.empty(10, np.int32)
    used = C.c_int32()
    nat.check(nat.lib.mir_compact_term_ids(nat.ptr(ids), len(ids), 10, nat.ptr(out), nat.ptr(remap), C.byref(used)))
    assert used.value == 5
    assert out.tolist() == [0, 1, 0, 2, 1, 3, 2, 2, 4]
    assert remap.tolist() == [3, -1, -1, 1, -1, 4, -1, 0, -1, 2]
    # in place, and against a numpy restatement on random data
    rng = np.random.default_rng(0)
    big = rng.integers(0, 5000, 200000).astype(np.int32)
    _, first = np.unique(big, return_index=True)
    order = big[np.sort(first)]
    want_map = np.full(5000, -1, np.int32)
    want_map[order] = np.arange(len(order), dtype=np.int32)
    buf = big.copy()
    remap = np.empty(5000, np.int32)
    nat.check(nat.lib.mir_compact_term_ids(nat.ptr(buf), len(buf), 5000, nat.ptr(buf), nat.ptr(remap), C.byref(used)))
    assert used.value == len(order)
    np.testing.assert_array_equal(remap, want_map)
    np.testing.assert_array_equal(buf, want_map[big])
    with pytest.raises(ValueError):
        bad, scratch = np.array([1, 12], np.int32), np.empty(10, np.int32)
        nat.check(nat.lib.mir_compact_term_ids(nat.ptr(bad), 2, 10, nat.ptr(bad), nat.ptr(scratch), C.byref(used)))
