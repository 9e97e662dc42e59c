"""Native Snowball-English stemmer (csrc/stem_english.cpp through the C ABI, host code - runs without a GPU)
against NLTK's SnowballStemmer("english"): the per-token step of keywords_preprocess (keywords_search.py:13-18).
Fixture: tests/golden/snowball_english.json.gz, 30k (token, stem) pairs written by make_snowball_fixture.py."""

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
    assert len(pairs) > 30000


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
    monkeypatch.setattr(ks, "_nltk_front_end", lambda: (frozenset({"the", "is", "of"}), str.split))
    assert ks.keywords_preprocess("The colours of the Alps , running") == ["the", "colour", "alp", ",", "run"]


def test_without_nltk_data_the_front_end_refuses():
    try:
        import nltk  # noqa: F401
        from nltk.corpus import stopwords

        stopwords.words("english")
    except (ImportError, LookupError):
        ks._nltk_front_end.cache_clear()
        with pytest.raises(ImportError, match="punkt"):
            ks.keywords_preprocess("some text")
