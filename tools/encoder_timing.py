"""Encoder throughput: chunks/s on synthetic token ids (SURVEY 8(d) encoder input)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.embeddings.embeddings import BgeEncoder
from oracle import encoder as oe
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
model = oe.make_model(layers=12, seed=0)
enc = BgeEncoder.from_state_dict(model.state_dict())
rng = np.random.default_rng(99)
lens = np.full(n, fixed) if fixed else np.clip(np.round(rng.normal(220, 60, n)), 8, 512).astype(int)
seqs = [rng.integers(999, 30522, L).astype(np.int32) for L in lens]  # arrays: list->array conversion is not what is measured
for _ in range(2): enc.encode_ids(seqs[:256])
t0 = time.perf_counter(); out = enc.encode_ids(seqs); dt = time.perf_counter() - t0
tok = int(lens.sum())
flops = sum(12 * (L * (2*384*1152 + 2*384*384 + 4*384*1536) + 4 * L * L * 384) for L in lens)
print(f"{n} seqs, {tok} tokens: {dt*1e3:.1f} ms -> {n/dt:.0f} chunks/s, {tok/dt/1e6:.2f} Mtok/s, {flops/dt/1e12:.1f} TFLOP/s (real tokens)")
