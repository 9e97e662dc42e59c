"""BM25Retriever.from_doc_records over a combination of documents: first sight (per-token Python, as the reference
does on every request) vs a new combination of documents whose token-id arrays are cached."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.bm25_retriever import BM25Retriever

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 400
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 250
rng = np.random.default_rng(777)
words = np.array([f"w{i}" for i in range(50000)], dtype=object)
p = 1.0 / np.arange(1, 50001) ** 1.07; p /= p.sum()

class Item:
    __slots__ = ("chunk_index", "tokenized_text")
    def __init__(self, i, t): self.chunk_index, self.tokenized_text = i, t
class Rec:
    def __init__(self):
        lens = np.clip(np.round(rng.normal(150, 40, chunks)), 1, 400).astype(int)
        toks = words[rng.choice(50000, int(lens.sum()), p=p)]
        cuts = np.cumsum(lens)[:-1]
        self.text_index = [Item(i, t.tolist()) for i, t in enumerate(np.split(toks, cuts))]

t0 = time.perf_counter(); recs = [Rec() for _ in range(n_docs)]
n_tok = sum(len(i.tokenized_text) for r in recs for i in r.text_index)
print(f"{n_docs} documents x {chunks} chunks, {n_tok/1e6:.1f}M tokens (generated in {time.perf_counter()-t0:.0f} s)", flush=True)
t0 = time.perf_counter(); r1 = BM25Retriever.from_doc_records(recs, k=7, preprocess=str.split); t_first = time.perf_counter() - t0
sub = [recs[i] for i in rng.permutation(n_docs)[: n_docs - 1]]
t0 = time.perf_counter(); r2 = BM25Retriever.from_doc_records(sub, k=7, preprocess=str.split); t_new = time.perf_counter() - t0
t0 = time.perf_counter(); r3 = BM25Retriever.from_doc_records(sub, k=7, preprocess=str.split); t_same = time.perf_counter() - t0
print(f"first sight {t_first*1e3:.0f} ms | new combination of known documents {t_new*1e3:.0f} ms | same combination again {t_same*1e3:.2f} ms", flush=True)
print(r2._get_relevant_documents("w17 w400 w9")[:3])
