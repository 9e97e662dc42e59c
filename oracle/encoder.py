"""Oracle: the encoder's arithmetic in float32 on the CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference reaches the model through langchain-community ->
sentence-transformers 3.3.1 -> transformers (aidial_rag/embeddings/
embeddings.py:52-66): a BertModel, CLS pooling, L2 normalisation.  None of that
wrapper stack is installed here, but `transformers.BertModel` (third-party,
present) is the same arithmetic; with seeded random weights of the
bge-small-en shape it is the oracle for the HIP kernels.  PARITY WITH THE REAL
MODEL IS UNPINNED: no bge-small-en weights exist offline, and the reference's
only pin (tests/test_retrievers.py:90-104, top-1 chunk for one query) needs
them.
"""

import numpy as np
import torch


def make_model(layers: int = 12, seed: int = 0, scale: float = 1.0):
    from transformers import BertConfig, BertModel

    cfg = BertConfig(hidden_size=384, num_hidden_layers=layers, num_attention_heads=12, intermediate_size=1536,
                     vocab_size=30522, max_position_embeddings=512)
    torch.manual_seed(seed)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    if scale != 1.0:  # random init (std 0.02) gives near-linear layers; larger weights exercise softmax / GELU / LN
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "LayerNorm" not in n and p.dim() == 2:
                    p.mul_(scale)
                elif "LayerNorm" in n or p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.05)
    return m


@torch.no_grad()
def hidden_states(model, sequences, layers_to_run=None):
    """Per-sequence float32 hidden states after `layers_to_run` layers (None = all), no padding involved."""
    outs = []
    for ids in sequences:
        t = torch.tensor([list(ids)], dtype=torch.long)
        o = model(input_ids=t, attention_mask=torch.ones_like(t), output_hidden_states=True)
        hs = o.hidden_states[-1 if layers_to_run is None else layers_to_run][0]
        outs.append(hs.numpy().astype(np.float32))
    return outs


def embed(model, sequences, normalize=True) -> np.ndarray:
    cls = np.stack([h[0] for h in hidden_states(model, sequences)])
    if normalize:
        cls = cls / np.maximum(np.linalg.norm(cls, axis=1, keepdims=True), 1e-12)
    return cls.astype(np.float32)
