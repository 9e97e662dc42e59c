"""One short query through the encoder, a few times: run under `rocprofv3 --kernel-trace` to see the launch timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.embeddings.embeddings import BgeEncoder
from oracle import encoder as oe
enc = BgeEncoder.from_state_dict(oe.make_model(layers=12, seed=0).state_dict())
rng = np.random.default_rng(5)
seqs = [rng.integers(999, 30522, 24).astype(np.int32)]
for _ in range(20): enc.encode_ids(seqs)
