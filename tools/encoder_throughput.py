"""Encoder throughput leg alone (8192 synthetic chunks, token-id level), for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from aidial_rag_amd.embeddings.embeddings import BgeEncoder
enc = BgeEncoder.from_state_dict(bench.random_bge_small_state_dict(np))
rng = np.random.default_rng(99)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lens = np.clip(np.round(rng.normal(220, 60, n)), 8, 512).astype(np.int64)
seqs = [rng.integers(999, 30522, L).astype(np.int32) for L in lens]
enc.encode_ids(seqs[:512])
for rep in range(3):
    t0 = time.perf_counter()
    enc.encode_ids(seqs)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt:.0f} chunks/s  ({dt * 1e3:.1f} ms)", flush=True)
