"""The native batched keywords_preprocess (csrc/keywords_preprocess.cpp; keywords_search.py:13-18 upstream, called per
chunk at bm25_retriever.py:30-39,106-114): its Treebank rules against the 24 744 sentences tokenised by nltk 3.6.5, and the
whole chain - sentence split, rules, stopword filter, str.lower(), Snowball - against the Python mirror text by text,
on prose and on adversarial strings (quotes, Unicode classes, final sigma, U+0130).  Host code: no GPU."""

import gzip
import json
import os
import random
import subprocess
import sys

import pytest

from aidial_rag_amd import keywords_search as ks


@pytest.fixture(scope="module")
def pairs(golden_dir):
    d = json.load(gzip.open(os.path.join(golden_dir, "treebank_tokenize.json.gz"), "rt", encoding="utf-8"))
    assert len(d["pairs"]) > 20000
    return d["pairs"]


def test_native_treebank_rules_equal_nltk_on_the_fixture(pairs):
    got = ks.treebank_tokenize_batch([s for s, _ in pairs])
    bad = [(s, g, t) for (s, t), g in zip(pairs, got) if g != t]
    assert not bad, bad[:5]


def test_native_chain_equals_the_python_mirror_on_prose(pairs):
    """Chunks made of the fixture's sentences (~1000 characters each, as the product's chunks): word_tokenize and the
    whole keywords_preprocess, native batch vs the Python mirror."""
    rng = random.Random(5)
    sents = [s for s, _ in pairs]
    chunks = []
    for _ in range(1500):
        t, n = "", rng.randint(200, 1200)
        while len(t) < n:
            t += rng.choice(sents) + rng.choice([" ", "  ", "\n", " \n ", ""])
        chunks.append(t)
    ks._front_end()
    if ks.FRONT_END != "restated":
        pytest.skip("NLTK data present: the product uses NLTK's tokenizer, not the native restatement")
    got_w = ks.word_tokenize_batch(chunks)
    got_k = ks.keywords_preprocess_batch(chunks)
    for c, gw, gk in zip(chunks, got_w, got_k):
        assert gw == ks._word_tokenize_restated(c), c
        assert gk == ks.keywords_preprocess(c), c
    assert sum(len(k) for k in got_k) > 100000


ADVERSARIAL_ALPHABET = (
    list("abcdefgimnorstuvwyzADEGILMNORSTVWYZ") * 3 + list("  \t\n\r\x0b\x0c\x1c\x1d\x85\xa0  　") + list(".,:;!?'\"`-*()[]{}<>@#$%&_/\\0123456789")
    + list("«»“”‘’„…éßİıKſΣσςΑᾳ­́٠१１中한\U0001d7ce\U00010400\U0001f600²½Ⅰ")
)
ADVERSARIAL_WORDS = ["cannot", "Cannot", "CANNOT", "d'ye", "gimme", "gonna", "gotta", "lemme", "more'n", "wanna", "'tis", "'twas", "'Tis", "can't", "won't",
                     "I'm", "they'll", "WE'RE", "it's", "isn't", "N'T", "'em", "''", "``", "--", "...", "Dr.", "e.g.", "U.S.", "Mr.", "J.", "etc.", "İstanbul",
                     "ΟΔΥΣΣΕΥΣ", "Σ", "AΣ.", "ſkis", "Kelvin", "generé", "1,000", "3:15", "a:b", "x,", "y:"]


def adversarial(rng, n):
    out = []
    for _ in range(n):
        parts = []
        for _ in range(rng.randint(1, 40)):
            if rng.random() < 0.35:
                parts.append(rng.choice(ADVERSARIAL_WORDS))
            else:
                parts.append("".join(rng.choice(ADVERSARIAL_ALPHABET) for _ in range(rng.randint(1, 6))))
            parts.append(rng.choice([" ", " ", " ", "", ". ", ", ", "\n", "! ", "? ", ".\" ", " \"", " '", "' "]))
        out.append("".join(parts))
    return out


def test_native_chain_equals_the_python_mirror_on_adversarial_strings():
    ks._front_end()
    if ks.FRONT_END != "restated":
        pytest.skip("NLTK data present")
    texts = adversarial(random.Random(11), 6000) + ["", " ", "\n", ".", "..", "'", "\"", "\"\"", "a.\n", "a,\n", "a:", ",\n", "x. \n\"", "(\"a\")", "''a", "a''b"]
    got_t = ks.treebank_tokenize_batch(texts)
    got_w = ks.word_tokenize_batch(texts)
    got_k = ks.keywords_preprocess_batch(texts, threads=3)
    for t, gt, gw, gk in zip(texts, got_t, got_w, got_k):
        assert gt == ks.treebank_tokenize(t), repr(t)
        assert gw == ks._word_tokenize_restated(t), repr(t)
        assert gk == ks.stem_tokens([x for x in ks._word_tokenize_restated(t) if x not in ks.ENGLISH_STOPWORDS]), repr(t)


def test_lazy_token_lists_behave_like_the_lists_and_map_through_the_vocabulary(pairs):
    """`BM25Retriever.build_index` hands out TokenList views (no str object per token): list-like for every reader of
    `tokenized_text`, pickled as plain lists, and mapped to term ids batch-wise - the same ids, in the same first-seen order,
    as token-by-token insertion gives (bm25_retriever.py:78 builds its dicts in that order)."""
    import asyncio
    import pickle

    import numpy as np

    from aidial_rag_amd.retrievers import bm25_retriever as br

    class Chunk:
        def __init__(self, t):
            self.text = t

    texts = [s for s, _ in pairs[:6000]] + ["", "   "]
    eager = ks.keywords_preprocess_batch(texts)
    items = asyncio.run(br.BM25Retriever.build_index([Chunk(t) for t in texts]))
    assert [i.chunk_index for i in items] == list(range(len(texts)))
    if ks.FRONT_END != "restated":
        pytest.skip("NLTK data present: lists, not views")
    for it, want in zip(items, eager):
        tl = it.tokenized_text
        assert isinstance(tl, ks.TokenList) and len(tl) == len(want) and tl == want and list(tl) == want
        if want:
            assert tl[0] == want[0] and tl[-1] == want[-1] and tl[1:3] == want[1:3]
    assert pickle.loads(pickle.dumps(items[5].tokenized_text)) == eager[5] and type(pickle.loads(pickle.dumps(items[5].tokenized_text))) is list
    assert br.BM25Retriever.has_index([type("D", (), {"text_index": items})()])

    class Item:
        def __init__(self, i, t):
            self.chunk_index, self.tokenized_text = i, t

    saved = dict(br._VOCAB)
    try:
        br._VOCAB.clear()
        (_, lens_e, ids_e), _ = br._doc_token_ids([Item(i, t) for i, t in enumerate(eager)])
        br._VOCAB.clear()
        (_, lens_l, ids_l), _ = br._doc_token_ids(items)
        np.testing.assert_array_equal(lens_l, lens_e)
        np.testing.assert_array_equal(ids_l, ids_e)
    finally:
        br._VOCAB.clear()
        br._VOCAB.update(saved)


def test_thread_count_does_not_change_the_result(pairs):
    texts = [s for s, _ in pairs[:3000]]
    one = ks.keywords_preprocess_batch(texts, threads=1)
    assert one == ks.keywords_preprocess_batch(texts, threads=8) == ks.keywords_preprocess_batch(texts)


def test_a_nul_character_is_refused_like_stem_tokens_refuses_it():
    with pytest.raises(ValueError):
        ks.keywords_preprocess_batch(["fine", "not\x00fine"])
    assert ks.keywords_preprocess_batch([]) == []


def test_unicode_tables_are_this_interpreters(tmp_path):
    """csrc/unicode_tables.h is generated from the running interpreter's str / re answers (tools/gen_unicode_tables.py);
    a stale header (other Python / Unicode version) would make the native path and the mirror disagree on rare code points."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(os.path.join(root, "ai-dial-rag_amd", "csrc", "unicode_tables.h")).read()
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_unicode_tables", os.path.join(root, "tools", "gen_unicode_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.OUT = str(tmp_path / "unicode_tables.h")
    gen.main()
    assert open(gen.OUT).read() == committed
