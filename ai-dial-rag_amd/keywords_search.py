"""Keyword preprocessing, same surface as aidial_rag/keywords_search.py:13-18.

``keywords_preprocess(text)`` = NLTK ``word_tokenize`` -> drop tokens found in the English stopword list
(compared BEFORE lower-casing, so "The" survives) -> Snowball-English stem of the lower-cased token.

What is native / restated here, and what it is pinned against:

* the stemmer: ``stem_tokens`` hands a batch of tokens to ``mir_stem_english`` (C++, csrc/stem_english.cpp), which
  follows NLTK's EnglishStemmer including its quirks - pinned token by token (tests/golden/snowball_english.json.gz).
  In the reference it is pure Python and, once the encoder runs on a GPU, the slowest step of the index build
  (bm25_retriever.py:30-39,112);
* the word tokenizer: ``treebank_tokenize`` restates NLTK's ``NLTKWordTokenizer`` (the per-sentence half of
  ``word_tokenize``: an ordered list of regular-expression substitutions, then a whitespace split) - pinned on 24 744
  sentences written by nltk 3.6.5 (tests/golden/treebank_tokenize.json.gz; the reference pins 3.9.1);
* the sentence splitter in FRONT of it is Punkt, a TRAINED model whose parameters (nltk_data `punkt`) are not in
  this image: ``split_sentences`` is a rule-based APPROXIMATION (sentence-final punctuation followed by whitespace and
  a capital / digit / opening quote, a short list of abbreviations excepted) and UNPINNED.  It only matters for the one
  thing sentence boundaries change: a period at the end of a non-final sentence becomes a token of its own.  A query
  is normally one sentence, where ``word_tokenize`` IS the Treebank tokenizer;
* the stopword list: NLTK's English list as shipped with nltk_data through 2023 (179 words), restated from memory and
  UNPINNED (the data file is absent; newer nltk_data releases extend the list).

When NLTK and its data ARE installed, ``keywords_preprocess`` uses them (exact by construction); otherwise the
restatements above.  Callers that hold token lists already (the stored ``tokenized_text`` of a DocumentRecord)
never need any of this.
"""

import ctypes as C
import logging
from functools import lru_cache
from typing import List, Sequence

LANG = "english"
_SEP = "\x00"

logger = logging.getLogger(__name__)


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


# ---- NLTKWordTokenizer (nltk/tokenize/destructive.py, nltk 3.6.5), restated: ordered (pattern, replacement) rules ----
import re

_STARTING_QUOTES = [
    (re.compile("([\u00ab\u201c\u2018\u201e]|[`]+)"), r" \1 "),
    (re.compile(r'^"'), r"``"),
    (re.compile(r"(``)"), r" \1 "),
    (re.compile(r"([ \(\[{<])(\"|'{2})"), r"\1 `` "),
    (re.compile(r"(?i)(')(?!re|ve|ll|m|t|s|d|n)(\w)\b"), r"\1 \2"),
]
_PUNCTUATION = [
    (re.compile("([^\\.])(\\.)([\\]\\)}>\"'\u00bb\u201d\u2019 ]*)\\s*$"), r"\1 \2 \3 "),
    (re.compile(r"([:,])([^\d])"), r" \1 \2"),
    (re.compile(r"([:,])$"), r" \1 "),
    (re.compile(r"\.{2,}"), r" \g<0> "),
    (re.compile(r"[;@#$%&]"), r" \g<0> "),
    (re.compile(r"([^\.])(\.)([\]\)}>\"']*)\s*$"), r"\1 \2\3 "),
    (re.compile(r"[?!]"), r" \g<0> "),
    (re.compile(r"([^'])' "), r"\1 ' "),
    (re.compile(r"[*]"), r" \g<0> "),
]
_PARENS_BRACKETS = (re.compile(r"[\]\[\(\)\{\}\<\>]"), r" \g<0> ")
_DOUBLE_DASHES = (re.compile(r"--"), r" -- ")
_ENDING_QUOTES = [
    (re.compile("([\u00bb\u201d\u2019])"), r" \1 "),
    (re.compile(r'"'), " '' "),
    (re.compile(r"(\S)('')"), r"\1 \2 "),
    (re.compile(r"([^' ])('[sS]|'[mM]|'[dD]|') "), r"\1 \2 "),
    (re.compile(r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r"\1 \2 "),
]
_CONTRACTIONS = [re.compile(p) for p in (
    r"(?i)\b(can)(?#X)(not)\b", r"(?i)\b(d)(?#X)('ye)\b", r"(?i)\b(gim)(?#X)(me)\b", r"(?i)\b(gon)(?#X)(na)\b",
    r"(?i)\b(got)(?#X)(ta)\b", r"(?i)\b(lem)(?#X)(me)\b", r"(?i)\b(more)(?#X)('n)\b", r"(?i)\b(wan)(?#X)(na)\s",
    r"(?i) ('t)(?#X)(is)\b", r"(?i) ('t)(?#X)(was)\b")]


def treebank_tokenize(sentence: str) -> List[str]:
    """``NLTKWordTokenizer().tokenize(sentence)``: the substitutions in NLTK's order, then a whitespace split."""
    text = sentence
    for rx, sub in _STARTING_QUOTES:
        text = rx.sub(sub, text)
    for rx, sub in _PUNCTUATION:
        text = rx.sub(sub, text)
    text = _PARENS_BRACKETS[0].sub(_PARENS_BRACKETS[1], text)
    text = _DOUBLE_DASHES[0].sub(_DOUBLE_DASHES[1], text)
    text = " " + text + " "
    for rx, sub in _ENDING_QUOTES:
        text = rx.sub(sub, text)
    for rx in _CONTRACTIONS:
        text = rx.sub(r" \1 \2 ", text)
    return text.split()


# ---- sentence splitting: an APPROXIMATION of Punkt (see the module docstring) ----
_ABBREVIATIONS = frozenset("""mr mrs ms dr prof sr jr st vs etc inc ltd co corp no fig figs eq vol pp ed eds e.g i.e cf al approx dept est
jan feb mar apr jun jul aug sep sept oct nov dec mon tue wed thu fri sat sun u.s u.k a.m p.m ph.d gen col lt sgt capt rev hon""".split())
_SENT_END = re.compile(r"""([.?!]+)(["'\u201d\u2019)\]]*)(\s+)(?=["'\u201c\u2018(\[]*[A-Z0-9])""")


def split_sentences(text: str) -> List[str]:
    out, start = [], 0
    for m in _SENT_END.finditer(text):
        if m.group(1) == ".":
            before = text[start : m.start()].rsplit(None, 1)
            last = (before[-1] if before else "").lower().strip("\"'([")
            if last in _ABBREVIATIONS or (len(last) == 1 and last.isalpha()):  # "Dr. Smith", initials: "J. Smith"
                continue
        end = m.end(2)
        out.append(text[start:end])
        start = m.end()
    tail = text[start:]
    if tail.strip():
        out.append(tail)
    return [s for s in out if s.strip()]


def _word_tokenize_restated(text: str) -> List[str]:
    return [tok for sent in split_sentences(text) for tok in treebank_tokenize(sent)]


ENGLISH_STOPWORDS = frozenset("""i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself yourselves he him his
himself she she's her hers herself it it's its itself they them their theirs themselves what which who whom this that that'll these those am
is are was were be been being have has had having do does did doing a an the and but if or because as until while of at by for with about
against between into through during before after above below to from up down in out on off over under again further then once here there when
where why how all any both each few more most other some such no nor not only own same so than too very s t can will just don don't should
should've now d ll m o re ve y ain aren aren't couldn couldn't didn didn't doesn doesn't hadn hadn't hasn hasn't haven haven't isn isn't ma
mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't wasn wasn't weren weren't won won't wouldn wouldn't""".split())


FRONT_END = None  # "nltk" | "restated" once _front_end() has chosen (front_end_info() reports it)
_warned_multi_sentence = False


@lru_cache(maxsize=1)
def _front_end():
    """(stopwords, word_tokenize): NLTK's own when it is installed WITH its data, else the restatements above - and then
    it SAYS so, once: the restated sentence splitter approximates Punkt (a trained model) and the stopword list is restated
    from memory, both unpinned, so tokens can differ from the reference's `tokenized_text`."""
    global FRONT_END
    try:
        from nltk.corpus import stopwords
        from nltk.tokenize import word_tokenize

        stop = frozenset(stopwords.words(LANG))
        word_tokenize("probe the tokenizer data")
        FRONT_END = "nltk"
        return stop, word_tokenize
    except (ImportError, LookupError) as e:
        FRONT_END = "restated"
        logger.warning("keywords_preprocess: NLTK or its punkt / stopwords data is not available (%s); using the restated Treebank "
                       "tokenizer (pinned), a rule-based sentence splitter that APPROXIMATES Punkt and a stopword list restated from "
                       "memory (both unpinned): tokens of multi-sentence texts can differ from the reference's.", type(e).__name__)
        return ENGLISH_STOPWORDS, _word_tokenize_restated


def front_end_info() -> dict:
    """Which tokenizer front end keywords_preprocess uses in this process, and what of it is pinned."""
    _front_end()
    if FRONT_END == "nltk":
        return {"front_end": "nltk", "word_tokenize": "nltk (Punkt + Treebank)", "stopwords": "nltk_data", "stemmer": "native Snowball (pinned)"}
    return {"front_end": "restated", "word_tokenize": "Treebank restated (pinned on 24 744 sentences) + rule-based sentence split (APPROXIMATES Punkt, unpinned)",
            "stopwords": "179-word list restated from memory (unpinned)", "stemmer": "native Snowball (pinned)"}


_MAYBE_MULTI = re.compile(r"[.?!][\"'\u201d\u2019)\]]*\s")  # cheap necessary condition of _SENT_END (not a full split per call)


def _note_multi_sentence(text: str) -> None:
    """The one place the approximation can change tokens: a text the restated splitter cuts into several sentences."""
    global _warned_multi_sentence
    if FRONT_END == "restated" and not _warned_multi_sentence and _MAYBE_MULTI.search(text) and len(split_sentences(text)) > 1:
        _warned_multi_sentence = True
        logger.warning("keywords_preprocess: tokenising multi-sentence text with the approximate sentence splitter (NLTK data absent); "
                       "sentence-final periods may be attached differently from the reference's word_tokenize.  Pass stored "
                       "`tokenized_text`, a `preprocess` callable, or install nltk with punkt + stopwords to remove the difference.")


def word_tokenize(text: str) -> List[str]:
    return _front_end()[1](text)


def _native_batch(texts: Sequence[str], mode: int, threads: int) -> List[List[str]]:
    """``mir_keywords_preprocess`` over a batch: one call, all host cores, the GIL released for its duration.  The Python
    side is what is left on one thread (a str object per token is the product's own format), so it is kept lean: one
    encode of the joined batch when it is pure ASCII, one decode + split per chunk on the way back."""
    import numpy as np

    from . import _native as nat

    n = len(texts)
    if n == 0:
        return []
    blob = "".join(texts).encode("utf-8", "surrogatepass")
    offsets = np.zeros(n + 1, dtype=np.int64)  # byte offsets: a text's character count when it is ASCII (the common case)
    np.cumsum([len(t) if t.isascii() else len(t.encode("utf-8", "surrogatepass")) for t in texts], out=offsets[1:])
    assert int(offsets[-1]) == len(blob)
    res = C.c_void_p()
    nat.check(nat.lib.mir_keywords_preprocess(blob, offsets.ctypes.data, n, threads, mode, C.byref(res)))
    try:
        ptr, nbytes, counts_p, ends_p, ntok = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_void_p(), C.c_int64()
        nat.check(nat.lib.mir_kwp_result_data(res, C.byref(ptr), C.byref(nbytes), C.byref(counts_p), C.byref(ends_p), C.byref(ntok)))
        raw = C.string_at(ptr.value, nbytes.value) if nbytes.value else b""
        ends = np.ctypeslib.as_array(C.cast(ends_p, C.POINTER(C.c_int64)), shape=(n,)).tolist()
    finally:
        nat.lib.mir_kwp_result_free(res)
    out, a = [], 0
    for b in ends:  # every token is FOLLOWED by a NUL: [a, b - 1) holds a text's tokens, NUL-separated
        out.append(raw[a : b - 1].decode("utf-8", "surrogatepass").split(_SEP) if b > a else [])
        a = b
    return out


class TokenBatch:
    """The tokens of one native batch, deduplicated: `uniq` = its distinct tokens in order of first appearance (ONE str object
    each), `ids` = every token as an index into them (int32, all texts back to back), `offsets[i] .. offsets[i + 1]` = text i."""

    __slots__ = ("uniq", "ids", "offsets", "_global", "_lock")

    def __init__(self, uniq: List[str], ids, offsets):
        import threading

        self.uniq, self.ids, self.offsets = uniq, ids, offsets
        self._global = None  # (vocabulary object, int32 table uniq index -> its term id): bm25_retriever._doc_token_ids
        self._lock = threading.Lock()

    def global_ids(self, vocab: dict, lock) -> "np.ndarray":
        """uniq index -> term id in `vocab` (new tokens get the next ids, in order of first appearance), once per batch."""
        import numpy as np

        with self._lock:
            g = self._global
            if g is None or g[0] is not vocab or len(vocab) < g[2]:  # (a vocabulary that shrank was cleared: tests do)
                with lock:
                    for t in self.uniq:  # (ids in order of first appearance, as token-by-token insertion gives them)
                        if t not in vocab:
                            vocab[t] = len(vocab)
                    table = np.fromiter(map(vocab.__getitem__, self.uniq), np.int32, len(self.uniq))
                self._global = g = (vocab, table, len(vocab))
            return g[1]


class TokenList(Sequence):
    """One text's `tokenized_text` as a view into its TokenBatch: behaves like the reference's List[str] (len, iteration,
    indexing, equality with lists, pickling as a plain list) without a str object per token until somebody asks for one -
    an index build that goes on to the device (BM25Retriever.from_doc_records) never does."""

    __slots__ = ("batch", "a", "b")

    def __init__(self, batch: TokenBatch, a: int, b: int):
        self.batch, self.a, self.b = batch, a, b

    def __len__(self):
        return self.b - self.a

    def tolist(self) -> List[str]:
        u = self.batch.uniq
        return [u[i] for i in self.batch.ids[self.a : self.b].tolist()]

    def __iter__(self):
        return iter(self.tolist())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.tolist()[i]
        n = self.b - self.a
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self.batch.uniq[int(self.batch.ids[self.a + i])]

    def __eq__(self, other):
        if isinstance(other, TokenList):
            return self.tolist() == other.tolist()
        return self.tolist() == other

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __repr__(self):
        return repr(self.tolist())

    def __reduce__(self):  # stored DocumentRecords hold plain lists (document_record.py:42-52)
        return (list, (self.tolist(),))


def _native_batch_lazy(texts: Sequence[str], threads: int) -> List[TokenList]:
    """keywords_preprocess of a batch as TokenList views: the native call, its dedupe, and ~1 us of Python per text."""
    import numpy as np

    from . import _native as nat

    n = len(texts)
    if n == 0:
        return []
    blob = "".join(texts).encode("utf-8", "surrogatepass")
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(t) if t.isascii() else len(t.encode("utf-8", "surrogatepass")) for t in texts], out=offsets[1:])
    assert int(offsets[-1]) == len(blob)
    res = C.c_void_p()
    nat.check(nat.lib.mir_keywords_preprocess(blob, offsets.ctypes.data, n, threads, 0, C.byref(res)))
    try:
        nat.check(nat.lib.mir_kwp_result_dedupe(res, threads))
        ptr, nbytes, counts_p, ends_p, ntok = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_void_p(), C.c_int64()
        nat.check(nat.lib.mir_kwp_result_data(res, C.byref(ptr), C.byref(nbytes), C.byref(counts_p), C.byref(ends_p), C.byref(ntok)))
        uptr, ubytes, nuniq, ids_p = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_void_p()
        nat.check(nat.lib.mir_kwp_result_unique(res, C.byref(uptr), C.byref(ubytes), C.byref(nuniq), C.byref(ids_p)))
        uniq = C.string_at(uptr.value, ubytes.value).decode("utf-8", "surrogatepass").split(_SEP) if ubytes.value else []
        if uniq:
            uniq.pop()  # every token is FOLLOWED by a NUL
        ids = np.ctypeslib.as_array(C.cast(ids_p, C.POINTER(C.c_int32)), shape=(max(ntok.value, 1),))[: ntok.value].copy()
        counts = np.ctypeslib.as_array(C.cast(counts_p, C.POINTER(C.c_int32)), shape=(n,))
        tok_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=tok_off[1:])
    finally:
        nat.lib.mir_kwp_result_free(res)
    batch = TokenBatch(uniq, ids, tok_off)
    o = tok_off.tolist()
    return [TokenList(batch, o[i], o[i + 1]) for i in range(n)]


def keywords_preprocess_batch(texts: Sequence[str], threads: int = 0, lazy: bool = False) -> List[Sequence[str]]:
    """``[keywords_preprocess(t) for t in texts]`` (keywords_search.py:13-18 per chunk, bm25_retriever.py:30-39,112) in ONE
    native call on all host cores (`threads` <= 0) - csrc/keywords_preprocess.cpp.  `lazy`: TokenList views (list-like,
    no str object per token) instead of lists.  With NLTK and its data installed the tokenizer and the stopword list are
    NLTK's (exact by construction; only the stemmer is native then)."""
    stop, tokenize = _front_end()
    if FRONT_END == "nltk":
        return [stem_tokens([t for t in tokenize(text) if t not in stop]) for text in texts]
    return _native_batch_lazy(texts, threads) if lazy else _native_batch(texts, 0, threads)


def word_tokenize_batch(texts: Sequence[str], threads: int = 0) -> List[List[str]]:
    """``[word_tokenize(t) for t in texts]`` of the RESTATED front end, natively."""
    return _native_batch(texts, 1, threads)


def treebank_tokenize_batch(sentences: Sequence[str], threads: int = 0) -> List[List[str]]:
    """``[NLTKWordTokenizer().tokenize(s) for s in sentences]`` natively (what the pinned fixture checks)."""
    return _native_batch(sentences, 2, threads)


def keywords_preprocess(text: str) -> List[str]:
    stop, tokenize = _front_end()
    _note_multi_sentence(text)
    return stem_tokens([t for t in tokenize(text) if t not in stop])
