"""GPU parity of the encoder (C ABI -> HIP MFMA kernels) against transformers.BertModel float32 with the same
seeded random weights.  Tolerance: float16 operands with float32 accumulation (the reference's own CUDA path is
float16, embeddings.py:43-48): hidden states within 3e-2 absolute of float32 after 12 layers (values are O(1)
after LayerNorm), embedding cosine >= 0.9995.  SURVEY 8(d): encoder parity is reported, not gated at 1e-4."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from aidial_rag_amd import _native
    from aidial_rag_amd.embeddings.embeddings import BgeEncoder
    from oracle import encoder as oe

    assert _native.device_count() >= 1
    model = oe.make_model(layers=12, seed=0, scale=2.5)
    enc = BgeEncoder.from_state_dict(model.state_dict())
    rng = np.random.default_rng(99)
    seqs = []
    for L in (1, 5, 31, 32, 33, 64, 100, 257, 512, 220, 8):
        ids = rng.integers(999, 30522, L).tolist()
        ids[0] = 101
        if L > 1:
            ids[-1] = 102
        seqs.append(ids)
    return model, enc, seqs, oe


def split_hidden(hidden, seqs):
    out, off = [], 0
    for s in seqs:
        out.append(hidden[off : off + len(s)])
        off += (len(s) + 31) // 32 * 32
    return out


@pytest.mark.parametrize("layers", [0, 1, 2, 12])
def test_hidden_states_vs_transformers(setup, layers):
    model, enc, seqs, oe = setup
    _, hidden = enc.debug_hidden(seqs, layers)
    want = oe.hidden_states(model, seqs, layers)
    tol = {0: 2e-3, 1: 1.5e-2, 2: 2e-2, 12: 3e-2}[layers]
    for got, w, s in zip(split_hidden(hidden, seqs), want, seqs):
        err = np.abs(got - w).max()
        assert err < tol, f"len {len(s)} layers {layers}: max abs err {err}"


def test_embeddings_cls_normalised(setup):
    model, enc, seqs, oe = setup
    got = enc.encode_ids(seqs)
    want = oe.embed(model, seqs)
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    cos = (got * want).sum(1)
    assert cos.min() > 0.9995, cos
    raw = enc.encode_ids(seqs, normalize=False)
    np.testing.assert_allclose(raw / np.linalg.norm(raw, axis=1, keepdims=True), got, atol=1e-5)


def test_batching_is_invariant(setup):
    """A sequence's embedding does not depend on what else is in the batch (no cross-sequence leakage, padding
    masked) NOR on which kernels served it: batches of at most 256 token tiles take the latency kernels
    (encoder_kernels.h section L), larger ones the throughput kernels, and both must round identically."""
    model, enc, seqs, oe = setup
    rng = np.random.default_rng(7)
    filler = [rng.integers(999, 30522, 512).tolist() for _ in range(20)]  # 45 + 320 tiles: throughput kernels
    together = enc.encode_ids(seqs + filler)[: len(seqs)]
    small_batch = enc.encode_ids(seqs)  # 45 tiles: latency kernels
    np.testing.assert_array_equal(together, small_batch)
    alone = np.stack([enc.encode_ids([s])[0] for s in seqs])
    np.testing.assert_array_equal(together, alone)
    rev = enc.encode_ids((seqs + filler)[::-1])[::-1][: len(seqs)]
    np.testing.assert_array_equal(together, rev)
    # and layer by layer (hidden states of every token, not only the pooled CLS row)
    for layers in (1, 12):
        _, big = enc.debug_hidden(seqs + filler, layers)
        _, small = enc.debug_hidden(seqs, layers)
        np.testing.assert_array_equal(big[: len(small)], small)


def test_argument_errors(setup):
    model, enc, seqs, oe = setup
    with pytest.raises(ValueError):
        enc.encode_ids([[101] * 513])
    with pytest.raises(ValueError):
        enc.encode_ids([[101, 40000, 102]])
    with pytest.raises(ValueError):
        enc.encode_ids([[]])
    assert enc.encode_ids([]).shape == (0, 384)


def test_concurrent_query_encodes_share_passes():
    """embed_query from many threads: identical vectors to sequential calls, served by fewer encoder passes."""
    import threading

    from aidial_rag_amd.embeddings.embeddings import BgeEncoder
    from oracle import encoder as oe

    class Tok:  # stand-in tokenizer: the test is about batching, not WordPiece
        def __call__(self, texts, **kw):
            return {"input_ids": [[101] + [1000 + (ord(c) % 500) for c in t][:60] + [102] for t in texts]}

    enc = BgeEncoder.from_state_dict(oe.make_model(layers=2, seed=1).state_dict(), tokenizer=Tok())
    qs = [f"question number {i} about retrieval" for i in range(48)]
    want = [enc.embed_query(q) for q in qs]
    gc = enc._query_commit()
    p0 = gc.passes
    got = [None] * 48

    def work(t):
        for i in range(t, 48, 12):
            got[i] = enc.embed_query(qs[i])

    th = [threading.Thread(target=work, args=(t,)) for t in range(12)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == want  # a sequence's embedding does not depend on what else is in the batch (bit-identical)
    assert gc.passes - p0 < 48
