"""Host logic of the index-build surface (embeddings.py:79-108, batched.py:35-53 upstream): outer batches of 128,
results in order, progress lines - with concurrent outer batches sharing encoder passes.  No GPU: the encoder's
forward (`encode_packed`, the only thing that crosses the C ABI) is replaced by a recording stand-in; the GPU run of
the same surface is tests/test_gpu_encoder.py::test_build_embeddings_through_the_product_surface."""

import asyncio
import io
import threading
import time

import numpy as np

from aidial_rag_amd.embeddings import embeddings as emb


class FakeTokenizer:
    def __call__(self, texts, add_special_tokens=True, truncation=True, max_length=512):
        return {"input_ids": [[101] + [1000 + (ord(c) % 50) for c in t][: max_length - 2] + [102] for t in texts]}


class RecordingEncoder(emb.BgeEncoder):
    def __init__(self):
        super().__init__(None, 12, FakeTokenizer(), 0)
        self.passes = []
        self.lock = threading.Lock()

    def encode_packed(self, flat, lens, normalize=True):
        assert flat.dtype == np.int32 and lens.dtype == np.int32 and int(lens.sum()) == flat.shape[0]
        with self.lock:
            self.passes.append(len(lens))
        time.sleep(0.02)  # a pass costs the same whatever its fill
        out = np.zeros((len(lens), emb.EMBEDDING_LENGTH), np.float32)
        at = 0
        for i, n in enumerate(lens):
            out[i, 0] = n
            out[i, 1] = int(flat[at : at + n].sum()) % 9973
            at += n
        return out

    def close(self):
        pass


def test_build_embeddings_keeps_the_contract_and_coalesces_passes():
    enc = RecordingEncoder()
    emb.set_bge_embedding_impl(enc)
    try:
        texts = [f"chunk number {i}\nwith a newline " + "x" * (i % 37) for i in range(1000)]
        stage = io.StringIO()
        out = asyncio.run(emb.build_embeddings(iter(texts), stage))
        assert len(out) == 1000 and all(o.dtype == np.float32 and o.shape == (384,) for o in out)
        tok = FakeTokenizer()
        want = tok([t.replace("\n", " ") for t in texts])["input_ids"]  # newline replacement of the langchain wrapper
        assert [int(o[0]) for o in out] == [len(s) for s in want]          # order preserved across shared passes
        assert [int(o[1]) for o in out] == [sum(s) % 9973 for s in want]
        assert sum(enc.passes) == 1000
        assert len(enc.passes) < 8, enc.passes  # 8 outer batches of 128 rode fewer, larger passes
        lines = [l for l in stage.getvalue().split("\n") if l.strip()]
        assert lines and "8/8" in lines[-1]  # tqdm-style progress per outer batch, as upstream
        # a single outer batch and the empty case
        assert len(asyncio.run(emb.build_embeddings(["a", "b"]))) == 2
        assert asyncio.run(emb.build_embeddings([])) == []
        # embed_documents / aembed_query keep their return types
        docs = asyncio.run(emb.bge_embedding.aembed_documents(["a b", "c"]))
        assert isinstance(docs, list) and isinstance(docs[0], list) and isinstance(docs[0][0], float)
        q = asyncio.run(emb.bge_embedding.aembed_query("what?"))
        assert isinstance(q, list) and len(q) == 384
        assert int(q[0]) == len(emb.BGE_QUERY_INSTRUCTION_EN) + len("what?") + 2  # instruction prefix + [CLS]/[SEP]
    finally:
        emb.set_bge_embedding_impl(None)


def test_concurrent_documents_share_passes():
    """load_documents indexes several documents at once (documents.py:326 upstream: asyncio.gather)."""
    enc = RecordingEncoder()
    emb.set_bge_embedding_impl(enc)
    try:
        async def many():
            return await asyncio.gather(*[emb.build_embeddings([f"d{d} c{i}" for i in range(200)]) for d in range(6)])

        outs = asyncio.run(many())
        assert [len(o) for o in outs] == [200] * 6
        assert sum(enc.passes) == 1200 and len(enc.passes) < 12  # 12 outer batches in fewer passes
    finally:
        emb.set_bge_embedding_impl(None)


def test_encode_packed_checks_its_lengths_before_the_c_abi():
    import pytest

    enc = emb.BgeEncoder(None, 12, None, 0)
    try:
        with pytest.raises(ValueError, match="lengths"):
            enc.encode_packed(np.zeros(3, np.int32), np.asarray([2], np.int32))
    finally:
        enc.close = lambda: None
