"""Keyword preprocessing, same surface as aidial_rag/keywords_search.py:13-18.

Host-side string work (NLTK word_tokenize -> stopword filter applied BEFORE
lower-casing -> Snowball-English stem of the lower-cased token).  NLTK and its
`punkt` / `stopwords` data are not part of this build's image; the function
imports them on first use and raises ImportError when they are missing.
Callers that hold token lists already (the stored `tokenized_text` of a
DocumentRecord) never need it.  A native tokenizer is SURVEY.md 8(f) item 3.
"""

from functools import lru_cache
from typing import List

LANG = "english"


@lru_cache(maxsize=1)
def _nltk_pipeline():
    try:
        from nltk.corpus import stopwords
        from nltk.stem.snowball import SnowballStemmer
        from nltk.tokenize import word_tokenize

        stop = stopwords.words(LANG)
        word_tokenize("probe the tokenizer data")
    except (ImportError, LookupError) as e:  # pragma: no cover - depends on the host image
        raise ImportError(
            "keywords_preprocess needs nltk with the 'punkt' and 'stopwords' data "
            "(as the reference does); pass pre-tokenised text or a `preprocess` callable instead"
        ) from e
    return SnowballStemmer(LANG), stop, word_tokenize


def keywords_preprocess(text: str) -> List[str]:
    stemmer, stop, word_tokenize = _nltk_pipeline()
    return [stemmer.stem(t.lower()) for t in word_tokenize(text) if t not in stop]
