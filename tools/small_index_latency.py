"""Latency of one search call (host API, synchronous) on small indexes - BASELINE config C1 is ~1k chunks.

Per call times are reported as median and mean over 200 calls: on the GPU box the process runs under a CPU quota, and a
multi-threaded numpy loop (the oracle timing below) can leave the process throttled for ~70 ms some time later, which
lands in ONE call of a following loop and dominates its mean (seen in the rocprofv3 trace as a single host-side gap
between two searches; the kernels of every call take the same time).  The oracle is therefore timed after all GPU loops.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from oracle import embeddings_index as oi

rng = np.random.default_rng(1)
cases = []
for n in (200, 1000, 5000, 20000):
    docs = rng.standard_normal((n, 384)).astype(np.float32)
    docs /= np.linalg.norm(docs, axis=1, keepdims=True)
    dev = DeviceIndex.from_host(docs)
    for B in (1, 16):
        q = rng.standard_normal((B, 384))
        for _ in range(20):
            dev.search(q, 10, "sqeuclidean_dist")
        t = np.empty(200)
        for i in range(200):
            t0 = time.perf_counter()
            out = dev.search(q, 10, "sqeuclidean_dist")
            t[i] = time.perf_counter() - t0
        print(f"n={n} B={B}: GPU median {np.median(t)*1e6:.0f} us, mean {t.mean()*1e6:.0f} us, max {t.max()*1e6:.0f} us per call, "
              f"flagged {int(np.count_nonzero(out[5]))}", flush=True)
    cases.append((n, docs, q))
    del dev
for n, docs, q in cases:
    t0 = time.perf_counter()
    for _ in range(20):
        oi.find_flat(q[0], docs, "sqeuclidean_dist", 10)
    print(f"n={n}: CPU oracle {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per query", flush=True)
