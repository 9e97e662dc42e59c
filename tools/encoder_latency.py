"""Latency of encoding ONE short query (the live path's aembed_query, embeddings.py:93-96) and small batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.embeddings.embeddings import BgeEncoder
from oracle import encoder as oe
model = oe.make_model(layers=12, seed=0)
enc = BgeEncoder.from_state_dict(model.state_dict())
rng = np.random.default_rng(5)
for n, L in ((1, 24), (1, 128), (8, 24), (16, 24), (32, 64), (64, 64), (128, 64), (256, 64), (512, 64)):
    seqs = [rng.integers(999, 30522, L).astype(np.int32) for _ in range(n)]
    for _ in range(10): enc.encode_ids(seqs)
    t0 = time.perf_counter()
    for _ in range(100): enc.encode_ids(seqs)
    print(f"{n} x {L} tokens: {(time.perf_counter()-t0)/100*1e6:.0f} us per call", flush=True)
