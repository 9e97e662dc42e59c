"""keywords_preprocess (keywords_search.py:13-18 upstream) without NLTK installed: the Treebank word tokenizer against
24 744 sentences tokenised by nltk 3.6.5 (tests/golden/treebank_tokenize.json.gz, written by make_treebank_fixture.py),
the stopword rule (compared BEFORE lower-casing) and the whole chain through the native stemmer.  Host code: no GPU."""

import gzip
import json
import os

from aidial_rag_amd import keywords_search as ks


def test_treebank_tokenizer_equals_nltk_on_the_fixture(golden_dir):
    d = json.load(gzip.open(os.path.join(golden_dir, "treebank_tokenize.json.gz"), "rt", encoding="utf-8"))
    assert len(d["pairs"]) > 20000
    bad = [(s, ks.treebank_tokenize(s), t) for s, t in d["pairs"] if ks.treebank_tokenize(s) != t]
    assert not bad, bad[:5]


def test_single_sentence_word_tokenize_is_the_treebank_tokenizer():
    # a query is one sentence: no Punkt decision is involved
    for q in ("what is the climate in the alps?", "Colle di Cadibona", "isn't it \"odd\" (really)", "cost: $3,000.50 -- or more..."):
        assert ks.split_sentences(q) == [q]
        assert ks._word_tokenize_restated(q) == ks.treebank_tokenize(q)


def test_sentence_boundaries_make_the_period_a_token():
    toks = ks._word_tokenize_restated("The Alps are high. Dr. Smith climbed them in 1999. Really!")
    assert toks == ["The", "Alps", "are", "high", ".", "Dr.", "Smith", "climbed", "them", "in", "1999", ".", "Really", "!"]
    assert ks.split_sentences("") == [] and ks._word_tokenize_restated("   ") == []


def test_keywords_preprocess_chain():
    assert len(ks.ENGLISH_STOPWORDS) == 179 and "the" in ks.ENGLISH_STOPWORDS and "The" not in ks.ENGLISH_STOPWORDS
    got = ks.keywords_preprocess("The running dogs weren't easily fooled by the generously national Alps.")
    # "The" survives the (case-sensitive) stopword filter and is stemmed lower-case; "the", "by", "were" are dropped;
    # Treebank splits "weren't" into "were" + "n't"; the final period is a token
    assert got == ["the", "run", "dog", "n't", "easili", "fool", "generous", "nation", "alp", "."]


def test_the_unpinned_front_end_says_so(caplog):
    """Without NLTK's data the mirror falls back to restatements of which the sentence splitter and the stopword list are
    unpinned: it must say so (once) and report which front end is active."""
    import logging

    from aidial_rag_amd import keywords_search as ks

    info = ks.front_end_info()
    assert info["front_end"] in ("nltk", "restated")
    if info["front_end"] == "restated":
        assert "unpinned" in info["word_tokenize"] and "unpinned" in info["stopwords"]
        ks._warned_multi_sentence = False
        with caplog.at_level(logging.WARNING, logger=ks.__name__):
            ks.keywords_preprocess("One sentence here. And a second one follows.")
            ks.keywords_preprocess("Again two. Sentences.")
        hits = [r for r in caplog.records if "approximate sentence splitter" in r.getMessage()]
        assert len(hits) == 1  # once per process
