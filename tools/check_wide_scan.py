"""Repeat-run parity check of the 128-query scan against the oracle (B = 70 on a small index): used while chasing an intermittent ordering bug; prints mismatching queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from oracle import embeddings_index as oi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nchk = int(sys.argv[2]) if len(sys.argv) > 2 else 70
metric = sys.argv[3] if len(sys.argv) > 3 else "sqeuclidean_dist"
rng = np.random.default_rng(3)
docs = rng.standard_normal((n, 384)).astype(np.float32); docs /= np.linalg.norm(docs, axis=1, keepdims=True)
qs = rng.standard_normal((70, 384)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
dev = DeviceIndex.from_host(docs)
for rep in range(3):
    _, _, rows, dist, cnt, flags = dev.search(qs, 10, metric)
    bad = 0
    for i in range(nchk):
        w, wd = oi.find_flat(qs[i], docs, metric, 10)
        if not np.array_equal(rows[i], w):
            bad += 1
            if bad <= 2: print(i, rows[i].tolist(), w.tolist())
    print("rep", rep, "flags", int(flags.sum()), "bad", bad, flush=True)
