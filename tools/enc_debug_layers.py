"""Per-sequence max |hidden - float32 transformers| after 1 layer on the latency path and (with filler) the throughput path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.embeddings.embeddings import BgeEncoder
from oracle import encoder as oe
model = oe.make_model(layers=12, seed=0, scale=2.5)
enc = BgeEncoder.from_state_dict(model.state_dict())
rng = np.random.default_rng(99)
seqs = []
for L in (1, 5, 31, 32, 33, 64, 65, 96, 100, 128, 257, 512, 220, 8):
    ids = rng.integers(999, 30522, L).tolist(); ids[0] = 101
    if L > 1: ids[-1] = 102
    seqs.append(ids)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
want = oe.hidden_states(model, seqs, layers)
def split(hidden, ss):
    out, off = [], 0
    for s in ss:
        out.append(hidden[off:off + len(s)]); off += (len(s) + 31) // 32 * 32
    return out
_, hid = enc.debug_hidden(seqs, layers)
print("latency path  :", " ".join(f"{len(s)}:{np.abs(g - w).max():.3g}" for g, w, s in zip(split(hid, seqs), want, seqs)), flush=True)
filler = [rng.integers(999, 30522, 512).tolist() for _ in range(20)]
_, hid2 = enc.debug_hidden(seqs + filler, layers)
print("throughput    :", " ".join(f"{len(s)}:{np.abs(g - w).max():.3g}" for g, w, s in zip(split(hid2, seqs + filler), want, seqs)), flush=True)
