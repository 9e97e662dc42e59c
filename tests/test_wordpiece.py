"""The native WordPiece tokenizer (mir_wordpiece_*, host code: runs without a GPU) against transformers' BertTokenizer -
the tokenizer sentence-transformers calls for the reference's encoder (embeddings.py:57-64 upstream) - on a synthetic
vocabulary (the real bge-small-en vocabulary is not available offline): identical ids for prose, punctuation, accents,
CJK, control characters, over-long words, unknown pieces, truncation at 512."""

import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    from transformers import AutoTokenizer

    from aidial_rag_amd.embeddings.wordpiece import WordPieceTokenizer

    d = str(tmp_path_factory.mktemp("vocab"))
    rng = np.random.default_rng(3)
    letters = "abcdefghijklmnopqrstuvwxyz"
    words = sorted({"".join(rng.choice(list(letters), rng.integers(1, 9))) for _ in range(4000)})
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += list(letters) + list("0123456789") + list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") + ["##" + c for c in letters + "0123456789"]
    vocab += words + ["##" + w for w in words[::3]] + ["the", "alps", "climate", "##s", "##ing", "cafe", "uber", "naive", "resume",
                                                          "中", "文", "—", "…", "§", "ß", "stra", "##ße", "i"]
    seen, uniq = set(), []
    for v in vocab:
        if v not in seen:
            seen.add(v)
            uniq.append(v)
    open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8").write("\n".join(uniq) + "\n")
    json.dump({"tokenizer_class": "BertTokenizer", "do_lower_case": True, "model_max_length": 512}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    ref = AutoTokenizer.from_pretrained(d)
    mine = WordPieceTokenizer.from_vocab_file(os.path.join(d, "vocab.txt"), fallback=ref)
    return ref, mine, words


def both(ref, mine, texts):
    want = ref(texts, add_special_tokens=True, truncation=True, max_length=512)["input_ids"]
    got = mine(texts, add_special_tokens=True, truncation=True, max_length=512)["input_ids"]
    for t, g, w in zip(texts, got, want):
        assert list(g) == list(w), f"{t!r}: {g} vs {w}"


def test_hand_picked_cases(toks):
    ref, mine, words = toks
    both(ref, mine, [
        "", " ", "The Alps' climate -- what is it?", "Hello, World!  (testing) [brackets] {braces} <a@b.c> 50% $3.50 a_b",
        "Café Über naïve résumé İstanbul", "Straße straße", "中文 and text中mixed", "em—dash and ellipsis… § 5",
        "tabs\tand\nnewlines\r\nand\x0bvertical\x0cformfeed", "zero\x00byte and � replacement and soft­hyphen",
        "nbsp thin space ideographic　space line sep", "x" * 100 + " " + "y" * 101 + " " + "z" * 99,
        "unknownqqqqzzzzjjjj " + words[5] + words[7] + " tail", "ALL CAPS WORDS", "a1b2c3 123456 3.14159",
        "combining é ä ñ", "ẞ capital sharp s", "ǆ digraph Ĳ ligature", "가나다 hangul",
    ])


def test_random_prose_and_truncation(toks):
    ref, mine, words = toks
    rng = np.random.default_rng(11)
    punct = list(",.;:!?-'\"()")
    texts = []
    for i in range(400):
        n = int(rng.integers(1, 700 if i % 10 == 0 else 150))
        parts = []
        for _ in range(n):
            w = str(rng.choice(words))
            r = rng.random()
            if r < 0.1:
                w = w.capitalize()
            elif r < 0.15:
                w = w + str(rng.choice(words))  # compound: continuation pieces or [UNK]
            elif r < 0.2:
                w = w + str(rng.choice(punct))
            parts.append(w)
        texts.append(" ".join(parts))
    both(ref, mine, texts)
    long = mine([" ".join(words[:600])])["input_ids"][0]
    assert len(long) == 512 and long[0] == ref.cls_token_id and long[-1] == ref.sep_token_id


def test_beyond_bmp_goes_to_the_fallback_or_raises(toks):
    from aidial_rag_amd.embeddings.wordpiece import WordPieceTokenizer

    ref, mine, words = toks
    both(ref, mine, ["emoji \U0001f600 here", "plain", "math \U0001d400 bold"])
    # literal special tokens in the text: the reference's tokenizer emits their ids (matched before normalisation); the
    # native one hands such texts back instead of spelling them out as '[', 'sep', ']'
    both(ref, mine, ["a [SEP] b [MASK] [CLS] [UNK] [PAD]", "[sep] lower case is ordinary text", "ends with [CLS]", "[MASK]",
                     "[SE P] [ SEP] [SEPP] are not special"])
    lone = WordPieceTokenizer(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a"])
    assert lone(["a a"])["input_ids"] == [[2, 4, 4, 3]]
    with pytest.raises(ValueError):
        lone(["\U0001f600"])
    with pytest.raises(ValueError):
        WordPieceTokenizer(["a", "b"])  # no [UNK] / [CLS] / [SEP]


def test_encode_packed_is_the_concatenation_of_encode_arrays(toks):
    """The layout mir_encoder_encode takes (all ids back to back + lengths), incl. rows that went through the fallback,
    truncated rows and the empty batch."""
    import numpy as np

    ref, mine, words = toks
    texts = ["plain words here", "emoji \U0001f600 here", "", " ".join(words[:700]), "x"]
    arrays = mine.encode_arrays(texts, 64)
    flat, lens = mine.encode_packed(texts, 64)
    assert flat.dtype == np.int32 and lens.dtype == np.int32 and flat.flags["C_CONTIGUOUS"]
    assert lens.tolist() == [len(a) for a in arrays] and max(lens) == 64
    assert flat.tolist() == [int(t) for a in arrays for t in a]
    f0, l0 = mine.encode_packed([], 64)
    assert f0.shape == (0,) and l0.shape == (0,)
