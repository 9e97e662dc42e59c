"""Diagnostic: very frequent query terms on the 1M-document corpus, checked against the CSR oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
from oracle import bm25 as ob
n, vocab = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 50_000
rng = np.random.default_rng(777)
lens = np.clip(np.round(rng.normal(150, 40, n)), 1, 400).astype(np.int64)
indptr = np.concatenate(([0], np.cumsum(lens)))
toks = np.minimum(rng.zipf(1.07, int(lens.sum())) - 1, vocab - 1).astype(np.int32)
dev = DeviceBM25.from_token_ids(indptr, toks, vocab)
o = ob.BM25OkapiCSR(indptr, toks, vocab)
print("df[:8]", o.df[:8], flush=True)
cases = [[[0]], [[1]], [[0, 1]], [[0, 1, 2, 3, 4, 5, 6, 7]], [[5, 9, 30], [2, 2], [49, 48, 47, 46]], [[int(t) for t in np.random.default_rng(s).integers(0, 50, 6)] for s in range(64)]]
for qs in cases:
    print("case", qs[:3], len(qs), flush=True)
    idx, sc, cnt = dev.search(qs, 10)
    for i, q in enumerate(qs[:8]):
        want = o.get_scores(q)
        top = ob.top_n_indexes(want, 10)
        ok = np.array_equal(idx[i], top) and np.array_equal(sc[i], want[top])
        print("  ", q, "ok" if ok else f"MISMATCH {idx[i]} vs {top}", flush=True)
