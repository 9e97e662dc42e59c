"""Keyword preprocessing, same surface as aidial_rag/keywords_search.py:13-18.

``keywords_preprocess(text)`` = NLTK ``word_tokenize`` -> drop tokens found in the English stopword list
(compared BEFORE lower-casing, so "The" survives) -> Snowball-English stem of the lower-cased token.

The stemmer is native: ``stem_tokens`` hands a whole batch of tokens to ``mir_stem_english`` (C++,
csrc/stem_english.cpp), which follows NLTK's EnglishStemmer including its quirks and is pinned against it
token by token (tests/golden/snowball_english.json).  In the reference it is pure Python and, once the
encoder runs on a GPU, the slowest step of the index build (bm25_retriever.py:30-39,112).

Tokenisation and the stopword list still come from NLTK and its ``punkt`` / ``stopwords`` data, which are
not part of this build's image (Punkt is a trained model; there is nothing here to check a restatement
against): ``keywords_preprocess`` imports them on first use and raises ImportError when they are missing.
Callers that hold token lists already (the stored ``tokenized_text`` of a DocumentRecord) never need it.
"""

import ctypes as C
from functools import lru_cache
from typing import List, Sequence

LANG = "english"
_SEP = "\x00"


def stem_tokens(tokens: Sequence[str]) -> List[str]:
    """``[SnowballStemmer("english").stem(t) for t in tokens]`` in one native call."""
    from . import _native as nat

    if not tokens:
        return []
    lowered = [t.lower() for t in tokens]  # SnowballStemmer.stem lower-cases first; Python's rules are the reference's
    if any(_SEP in t for t in lowered):
        raise ValueError("a token contains a NUL character")
    buf = _SEP.join(lowered).encode("utf-8", "surrogatepass")
    out = C.create_string_buffer(len(buf) + 1)
    n = C.c_int64()
    nat.check(nat.lib.mir_stem_english(buf, len(buf), _SEP.encode(), out, C.byref(n)))
    return out.raw[: n.value].decode("utf-8", "surrogatepass").split(_SEP)


@lru_cache(maxsize=1)
def _nltk_front_end():
    try:
        from nltk.corpus import stopwords
        from nltk.tokenize import word_tokenize

        stop = frozenset(stopwords.words(LANG))
        word_tokenize("probe the tokenizer data")
    except (ImportError, LookupError) as e:  # pragma: no cover - depends on the host image
        raise ImportError(
            "keywords_preprocess needs nltk with the 'punkt' and 'stopwords' data "
            "(as the reference does); pass pre-tokenised text or a `preprocess` callable instead"
        ) from e
    return stop, word_tokenize


def keywords_preprocess(text: str) -> List[str]:
    stop, word_tokenize = _nltk_front_end()
    return stem_tokens([t for t in word_tokenize(text) if t not in stop])
