"""BM25 at config C3 (1M docs, 50k vocabulary, 4096-query batches of the SURVEY 8(d) mix) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
dev0 = torch.device("cuda", 0)
indptr, toks = bench.gen_bm25_corpus(np, torch, dev0, 1_000_000, 777)
dev = DeviceBM25.from_token_ids(indptr, toks, bench.BM25_VOCAB)
qs = bench.bm25_queries(np, 4096, 778 + 4096)
for _ in range(3):
    dev.search(qs, 10)
print("done", dev.info()["n_postings"])
