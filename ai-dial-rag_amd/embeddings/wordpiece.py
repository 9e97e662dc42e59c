"""Native WordPiece tokenizer for the BGE encoder (``mir_wordpiece_*``, csrc/wordpiece.cpp).

The reference tokenises inside sentence-transformers with the model's ``BertTokenizer`` (tokenizers' Rust
implementation: BertNormalizer -> BertPreTokenizer -> WordPiece; aidial_rag/embeddings/embeddings.py:57-64,79-96).
This class follows it rule for rule and is pinned against ``transformers.BertTokenizer`` on a synthetic vocabulary
(tests/test_wordpiece.py; the real bge-small-en vocabulary is not available offline).  It is a drop-in for the
``tokenizer`` argument of ``BgeEncoder``: ``tok(texts, add_special_tokens=True, truncation=True, max_length=512)``
-> ``{"input_ids": [...]}``.

The Unicode facts live on THIS side of the C ABI: per code point of the Basic Multilingual Plane, its class
(whitespace / removed / punctuation / CJK) and its normalised form (NFD, combining marks dropped, lower-cased), filled
from ``unicodedata`` once.  A text with a code point beyond the BMP goes to ``fallback`` (a reference tokenizer with
the same call signature) or raises.
"""

import ctypes as C
import threading
import unicodedata
from typing import List, Optional, Sequence

import numpy as np

from .. import _native as nat

_CLS_OTHER, _CLS_SPACE, _CLS_REMOVE, _CLS_PUNCT, _CLS_CJK, _CLS_FALLBACK = range(6)
_TABLES = None
_TABLES_LOCK = threading.Lock()


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:  # ASCII punctuation incl. $ + < = > ^ ` | ~
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return 0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0xF900 <= cp <= 0xFAFF  # the BMP ranges of BertNormalizer


def _tables(lowercase: bool):
    """(cls, ncls, map, maplen) over the BMP, as BertNormalizer(clean_text, handle_chinese_chars, strip_accents =
    lowercase, lowercase) treats every code point."""
    global _TABLES
    with _TABLES_LOCK:
        if _TABLES is not None and _TABLES[0] == lowercase:
            return _TABLES[1]
        cls = np.zeros(65536, np.uint8)
        ncls = np.zeros(65536, np.uint8)
        mp = np.zeros((65536, 3), np.uint32)
        ml = np.zeros(65536, np.uint8)
        for cp in range(65536):
            if 0xD800 <= cp <= 0xDFFF:  # surrogates cannot occur in valid UTF-8
                cls[cp] = _CLS_FALLBACK
                continue
            ch = chr(cp)
            cat = unicodedata.category(ch)
            if ch in "\t\n\r":
                k = _CLS_SPACE
            elif cp == 0 or cp == 0xFFFD or cat.startswith("C"):
                k = _CLS_REMOVE
            elif ch.isspace():
                k = _CLS_SPACE
            elif _is_cjk(cp):
                k = _CLS_CJK
            else:
                k = _CLS_OTHER  # (punctuation is split on the NORMALISED text: ncls)
            cls[cp] = k
            ncls[cp] = _CLS_SPACE if (ch in " \t\n\r" or ch.isspace()) else _CLS_PUNCT if _is_punct(ch) else _CLS_OTHER
            if k in (_CLS_OTHER, _CLS_CJK):
                s = ch
                if lowercase:  # strip_accents follows lowercase: NFD, drop Mn, then lower-case char by char
                    s = "".join(c for c in unicodedata.normalize("NFD", s) if unicodedata.category(c) != "Mn")
                    s = "".join(c.lower() for c in s)
                if len(s) > 3 or any(ord(c) > 0xFFFF for c in s):
                    cls[cp] = _CLS_FALLBACK
                else:
                    ml[cp] = len(s)
                    for j, c in enumerate(s):
                        mp[cp, j] = ord(c)
        _TABLES = (lowercase, (cls, ncls, np.ascontiguousarray(mp), ml))
        return _TABLES[1]


class WordPieceTokenizer:
    """``BertTokenizer`` behaviour over a vocab.txt, natively and on all host cores."""

    def __init__(self, vocab_lines: Sequence[str], do_lower_case: bool = True, fallback=None, threads: int = 0):
        self.vocab_size = len(vocab_lines)
        blob = "\n".join(vocab_lines).encode("utf-8")
        cls, ncls, mp, ml = _tables(do_lower_case)
        h = C.c_void_p()
        nat.check(nat.lib.mir_wordpiece_create(blob, len(blob), nat.ptr(cls), nat.ptr(ncls), nat.ptr(mp), nat.ptr(ml), 100, C.byref(h)))
        self._h, self.fallback, self.threads = h, fallback, threads

    @classmethod
    def from_vocab_file(cls, path: str, do_lower_case: bool = True, fallback=None) -> "WordPieceTokenizer":
        with open(path, encoding="utf-8") as f:
            lines = f.read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
        return cls(lines, do_lower_case, fallback)

    def __call__(self, texts: Sequence[str], add_special_tokens: bool = True, truncation: bool = True, max_length: int = 512):
        """The Hugging Face call shape: ``{"input_ids": [list of int, ...]}``."""
        if not add_special_tokens or not truncation:
            raise NotImplementedError("the encoder path tokenises with special tokens and truncation (sentence-transformers does)")
        return {"input_ids": [a.tolist() for a in self.encode_arrays(texts, max_length)]}

    def _encode_matrix(self, texts: List[str], max_length: int):
        """(ids [n, max_length] int32, lens [n] int32); rows the native tokenizer cannot do go through the fallback."""
        n = len(texts)
        enc = [t.encode("utf-8", "surrogatepass") for t in texts]
        ptr = np.zeros(n + 1, np.int64)
        np.cumsum([len(e) for e in enc], out=ptr[1:])
        blob = b"".join(enc)
        ids = np.zeros((n, max_length), np.int32)
        lens = np.zeros(n, np.int32)
        fb = np.zeros(n, np.uint8)
        nat.check(nat.lib.mir_wordpiece_encode(self._h, blob, nat.ptr(ptr), n, max_length, self.threads, nat.ptr(ids), nat.ptr(lens), nat.ptr(fb)))
        todo = np.flatnonzero(fb)
        if len(todo):
            if self.fallback is None:
                raise ValueError(f"text {int(todo[0])} holds a code point beyond the Basic Multilingual Plane (or invalid UTF-8) "
                                 "and no fallback tokenizer was given")
            got = self.fallback([texts[i] for i in todo], add_special_tokens=True, truncation=True, max_length=max_length)["input_ids"]
            for i, g in zip(todo, got):
                ids[int(i), : len(g)] = np.asarray(g, dtype=np.int32)
                lens[int(i)] = len(g)
        return ids, lens

    def encode_arrays(self, texts: Sequence[str], max_length: int = 512) -> List[np.ndarray]:
        """The same ids as int32 arrays (views of one matrix): what ``BgeEncoder`` feeds to ``mir_encoder_encode`` -
        building 8192 Python lists of ~220 ints only to turn them back into arrays cost more than the GPU pass."""
        texts = list(texts)
        if not texts:
            return []
        ids, lens = self._encode_matrix(texts, max_length)
        return [ids[i, : lens[i]] for i in range(len(texts))]

    def encode_packed(self, texts: Sequence[str], max_length: int = 512):
        """(flat ids int32, lens int32): the layout ``mir_encoder_encode`` takes, without a Python object per text - an
        outer batch travels through the group commit of ``BgeEncoder.embed_documents_numpy`` as these two arrays."""
        texts = list(texts)
        if not texts:
            return np.zeros(0, np.int32), np.zeros(0, np.int32)
        ids, lens = self._encode_matrix(texts, max_length)
        return np.ascontiguousarray(ids[np.arange(max_length, dtype=np.int32)[None, :] < lens[:, None]]), lens

    def close(self):
        if self._h:
            nat.lib.mir_wordpiece_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
