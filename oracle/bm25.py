"""Oracle: BM25Okapi scoring and the reference's top-n rule.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED upstream.  The arithmetic is NOT in the reference tree: it is
the third-party package ``rank-bm25==0.2.2`` (pyproject.toml:28,
poetry.lock:5283-5284), absent from this environment.  This file restates its
published ``BM25Okapi`` algorithm (k1=1.5, b=0.75, epsilon=0.25) and anchors on
the reference's own call sites:

* construction  - aidial_rag/retrievers/bm25_retriever.py:64-79
  (`BM25Okapi(tokenized_texts)`, ValueError("Text index is empty.") first)
* scoring       - bm25_retriever.py:81-84 (`get_scores`, then
  ``np.argsort(scores, kind="stable")[::-1][:n]``: ties, including the
  all-zero scores, go to the HIGHEST flat index)

The only rank-level pin upstream (tests/test_retrievers.py:82-88) needs
`unstructured` + NLTK data and cannot run here.

Tokens may be any hashable (str in the product, int term ids in benchmarks).
All arithmetic is float64 in the operation order of the package, so a GPU
implementation can be compared bit-for-bit.
"""

import math
from typing import Dict, Hashable, List, Sequence

import numpy as np


class BM25Okapi:
    def __init__(self, corpus: Sequence[Sequence[Hashable]], k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25):
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.corpus_size = 0
        self.doc_freqs: List[Dict[Hashable, int]] = []
        self.doc_len: List[int] = []
        self.idf: Dict[Hashable, float] = {}

        # BM25._initialize: per-document term frequencies, document
        # frequencies nd[word] in order of first appearance, average length.
        nd: Dict[Hashable, int] = {}
        total_len = 0
        for document in corpus:
            self.doc_len.append(len(document))
            total_len += len(document)
            freqs: Dict[Hashable, int] = {}
            for word in document:
                freqs[word] = freqs.get(word, 0) + 1
            self.doc_freqs.append(freqs)
            for word in freqs:
                nd[word] = nd.get(word, 0) + 1
            self.corpus_size += 1
        self.avgdl = total_len / self.corpus_size

        # BM25Okapi._calc_idf: idf = ln(N - n + 0.5) - ln(n + 0.5); terms with
        # negative idf are floored to epsilon * average_idf, where the average
        # runs over ALL terms (negative ones included), in insertion order.
        idf_sum = 0.0
        negative = []
        for word, freq in nd.items():
            idf = math.log(self.corpus_size - freq + 0.5) - math.log(freq + 0.5)
            self.idf[word] = idf
            idf_sum += idf
            if idf < 0:
                negative.append(word)
        self.average_idf = idf_sum / len(self.idf)
        eps = self.epsilon * self.average_idf
        for word in negative:
            self.idf[word] = eps

    def get_scores(self, query: Sequence[Hashable]) -> np.ndarray:
        """One pass per query token (repeats count again), all float64."""
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            q_freq = np.array([(doc.get(q) or 0) for doc in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (
                q_freq * (self.k1 + 1) / (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl))
            )
        return score


def top_n_indexes(scores: np.ndarray, n: int) -> np.ndarray:
    """bm25_retriever.py:81-84: stable ascending argsort, reversed, first n."""
    return np.argsort(scores, kind="stable")[::-1][:n]


def build(tokenized_texts: Sequence[Sequence[Hashable]]) -> BM25Okapi:
    """bm25_retriever.py:72-78: the guard, then the model."""
    if sum(map(len, tokenized_texts)) == 0:
        raise ValueError("Text index is empty.")
    return BM25Okapi(tokenized_texts)


# ---- vectorised restatement for large synthetic corpora --------------------
# Same arithmetic, same operation order per element, on CSR postings of
# integer term ids; used where the dict loop above is impractical (N >= 1e5).
# Checked against the dict loop in tests/test_oracle_bm25.py.


def group_postings(indptr: np.ndarray, term_ids: np.ndarray, vocab: int):
    """Integer bookkeeping only (no float arithmetic): the distinct (term, doc) pairs of a token stream, grouped by
    term with documents ascending, their term frequencies, document frequencies and the position of every term's
    first token (= insertion order of rank-bm25's `nd` dict).  -> (t_doc i64[P], t_tf i64[P], df i64[V], first_pos i64[V]).
    Tests at sizes where this numpy sort is too slow compute the same four arrays with a device sort and pass them as
    `grouped=` (checked against this function at small sizes, tests/test_oracle_bm25_fusion.py)."""
    n = len(indptr) - 1
    doc_len = np.diff(indptr).astype(np.int64)
    term_ids = np.asarray(term_ids)
    key = term_ids.astype(np.int64) * np.int64(max(n, 1)) + np.repeat(np.arange(n, dtype=np.int64), doc_len)
    key.sort()
    if len(key):
        flag = np.empty(len(key), dtype=bool)
        flag[0] = True
        np.not_equal(key[1:], key[:-1], out=flag[1:])
        starts = np.flatnonzero(flag)
        ukey = key[starts]
        t_tf = np.diff(np.append(starts, len(key))).astype(np.int64)
    else:
        ukey, t_tf = key, np.zeros(0, np.int64)
    t_doc = ukey % np.int64(max(n, 1))
    df = np.bincount(ukey // np.int64(max(n, 1)), minlength=vocab).astype(np.int64)
    first_pos = np.full(vocab, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first_pos, term_ids.astype(np.int64), np.arange(len(term_ids), dtype=np.int64))
    return t_doc, t_tf, df, first_pos


class BM25OkapiCSR:
    def __init__(self, indptr: np.ndarray, term_ids: np.ndarray, vocab: int, k1=1.5, b=0.75, epsilon=0.25, grouped=None):
        """`indptr[i]:indptr[i+1]` slices the TOKENS (with repeats, in text order) of doc i.  `grouped`: the result of
        `group_postings` computed elsewhere (integer bookkeeping; every float64 operation stays below)."""
        self.k1, self.b, self.epsilon = k1, b, epsilon
        n = len(indptr) - 1
        self.corpus_size = n
        self.doc_len = np.diff(indptr).astype(np.int64)
        self.avgdl = int(self.doc_len.sum()) / n
        self.t_doc, self.t_tf, df, first_pos = grouped if grouped is not None else group_postings(indptr, term_ids, vocab)
        # insertion order of nd = order of first appearance in the token stream
        present = np.flatnonzero(df)
        present = present[np.argsort(first_pos[present], kind="stable")]
        idf = np.zeros(vocab, dtype=np.float64)
        idf_sum = 0.0
        for t in present:
            v = math.log(n - int(df[t]) + 0.5) - math.log(int(df[t]) + 0.5)
            idf[t] = v
            idf_sum += v
        self.average_idf = idf_sum / len(present)
        idf[(idf < 0) & (df > 0)] = self.epsilon * self.average_idf
        self.idf = idf
        self.df = df
        self.t_ptr = np.concatenate(([0], np.cumsum(df)))

    def get_scores(self, query: Sequence[int]) -> np.ndarray:
        score = np.zeros(self.corpus_size)
        vocab = len(self.idf)
        for q in query:
            if q < 0 or q >= vocab or self.df[q] == 0:
                continue  # `(idf.get(q) or 0)` and `(doc.get(q) or 0)` are 0: adds +0.0
            sl = slice(self.t_ptr[q], self.t_ptr[q + 1])
            docs = self.t_doc[sl]
            tf = self.t_tf[sl]
            dl = self.doc_len[docs]
            score[docs] += self.idf[q] * (tf * (self.k1 + 1) / (tf + self.k1 * (1 - self.b + self.b * dl / self.avgdl)))
        return score
