"""Latency of one search call (host API, synchronous) on small indexes - BASELINE config C1 is ~1k chunks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from oracle import embeddings_index as oi
rng = np.random.default_rng(1)
for n in (200, 1000, 5000, 20000):
    docs = rng.standard_normal((n, 384)).astype(np.float32); docs /= np.linalg.norm(docs, axis=1, keepdims=True)
    dev = DeviceIndex.from_host(docs)
    for B in (1, 16):
        q = rng.standard_normal((B, 384))
        for _ in range(20): dev.search(q, 10, "sqeuclidean_dist")
        t0 = time.perf_counter()
        for _ in range(200): out = dev.search(q, 10, "sqeuclidean_dist")
        dt = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter()
        for _ in range(20): oi.find_flat(q[0], docs, "sqeuclidean_dist", 10)
        dc = (time.perf_counter() - t0) / 20
        print(f"n={n} B={B}: GPU {dt*1e6:.0f} us per call, CPU oracle {dc*1e6:.0f} us per query", flush=True)
