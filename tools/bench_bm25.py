"""BM25 at BASELINE config 3: 1M chunks, 50k-term vocabulary (SURVEY 8(d) corpus), one MI355X.
Reports build time, QPS at several batch sizes, postings bytes per query, and a CPU-oracle sample."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
from oracle import bm25 as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
vocab = 50_000
rng = np.random.default_rng(777)
lens = np.clip(np.round(rng.normal(150, 40, n)), 1, 400).astype(np.int64)
lens[rng.random(n) < 0.001] = 0
indptr = np.concatenate(([0], np.cumsum(lens)))
toks = np.minimum(rng.zipf(1.07, int(lens.sum())) - 1, vocab - 1).astype(np.int32)
t0 = time.perf_counter(); dev = DeviceBM25.from_token_ids(indptr, toks, vocab); t_build = time.perf_counter() - t0
info = dev.info()
qr = np.random.default_rng(778)
def make_queries(nq):
    out = []
    for i in range(nq):
        L = int(qr.integers(2, 9))
        q = qr.integers(50, 5000, L) if i % 2 else qr.integers(0, vocab, L)
        q = [int(t) for t in q]
        if i % 20 == 3: q[0] = vocab + 7
        if i % 20 == 5: q.append(q[0])
        out.append(q)
    return out
res = {"n_docs": n, "vocab": vocab, "postings": info["n_postings"], "hbm_bytes": info["hbm_bytes"], "build_s": round(t_build, 2), "runs": []}
o = ob.BM25OkapiCSR(indptr, toks, vocab)
df = o.df
for B in (1, 64, 512, 4096):
    qs = make_queries(B)
    dev.search(qs, 10)
    reps = max(1, 2048 // B)
    t0 = time.perf_counter()
    for _ in range(reps): idx, sc, cnt = dev.search(qs, 10)
    dt = (time.perf_counter() - t0) / reps
    touched = float(np.mean([sum(int(df[t]) for t in q if 0 <= t < vocab) for q in qs]))
    res["runs"].append({"batch": B, "qps": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "postings_touched_per_query": touched,
                        "algorithmic_bytes_per_query": 12 * touched, "achieved_GBps": round(12 * touched * B / dt / 1e9, 1)})
# query classes at B=4096: rare terms (rank >= 5000), mid band (50..5000), very frequent (rank < 50)
def make_class(kind, nq):
    lo, hi = {"rare": (5000, vocab), "mid": (50, 5000), "frequent": (0, 50)}[kind]
    return [[int(t) for t in qr.integers(lo, hi, int(qr.integers(2, 9)))] for _ in range(nq)]
res["classes"] = []
for kind in ("rare", "mid", "frequent"):
    qs = make_class(kind, 4096)
    dev.search(qs, 10)
    t0 = time.perf_counter()
    for _ in range(3): dev.search(qs, 10)
    dt = (time.perf_counter() - t0) / 3
    touched = float(np.mean([sum(int(df[t]) for t in q) for q in qs]))
    res["classes"].append({"class": kind, "batch": 4096, "qps": round(4096 / dt, 1), "postings_touched_per_query": touched,
                           "achieved_GBps": round(12 * touched * 4096 / dt / 1e9, 1)})
# parity + CPU baseline on the same corpus (vectorised restatement; the reference's dict loop is ~1000x slower)
qs = make_queries(32)
t0 = time.perf_counter(); want = [o.get_scores(q) for q in qs]; dt = time.perf_counter() - t0
idx, sc, cnt = dev.search(qs, 10)
ok = all(np.array_equal(idx[i], ob.top_n_indexes(want[i], 10)) and np.array_equal(sc[i], want[i][idx[i]]) for i in range(32))
res["cpu_oracle_csr_qps_1core"] = round(32 / dt, 2)
res["gpu_topn_and_scores_bit_identical_on_32_queries"] = bool(ok)
print(json.dumps(res))
