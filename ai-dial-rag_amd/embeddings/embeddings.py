"""Local text encoder, same surface as aidial_rag/embeddings/embeddings.py:52-108.

``bge_embedding.embed_documents / embed_query / aembed_documents_numpy /
aembed_query`` and ``build_embeddings`` keep their names and return types
(``List[np.ndarray float32[384]]`` for documents, ``List[float]`` for a query).
The forward pass is ``mir_encoder_encode`` (HIP MFMA kernels); tokenisation and
the text conventions of the wrappers the reference goes through stay on the
host:

* langchain-community 0.3.20 ``HuggingFaceBgeEmbeddings``: newlines are replaced
  by spaces; queries get the English BGE instruction prefix; documents get no
  prefix (embeddings.py:57-64,81,95).
* sentence-transformers 3.3.1: truncation at the model's 512 tokens, CLS pooling,
  ``normalize_embeddings=True``.

Weights: a Hugging Face BERT state dict (``epam/bge-small-en`` layout).  No
weights ship with this repository; ``BgeEncoder.from_state_dict`` /
``from_pretrained_dir`` load what the caller provides.  Model path and device
come from the same environment variables as upstream
(``BGE_EMBEDDINGS_MODEL_PATH``, ``BGE_EMBEDDINGS_DEVICE``, embeddings.py:30-36).
"""

import asyncio
import ctypes as C
import os
import threading
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

from .. import _native as nat
from .detect_device import DeviceType, detect_device

EMBEDDINGS_BATCH_SIZE = 128  # embeddings.py:24-26 (outer batch; the GPU path packs all of it into token tiles)
EMBEDDING_LENGTH = 384       # embeddings.py:69 finds this by encoding ""; bge-small-en is 384 by construction
BGE_QUERY_INSTRUCTION_EN = "Represent this question for searching relevant passages: "
MAX_TOKENS = 512
DOC_BATCHES_PER_PASS = 64    # outer batches one shared encode may carry (8192 chunks ~ 18 full passes of the kernels)
# outer batches `build_embeddings` keeps in flight (their tokenisation runs in threads).  While one shared encode carries
# 20-30 of them, as many again must be tokenising for the next one: 32 -> 64 is 54k -> 57.5k chunks/s (128: 56.5k).  A leader
# that waits a few ms for batches still being tokenised (instead of leaving with the first alone) was measured too: no gain.
BUILD_IN_FLIGHT = 64
# `build_embeddings` over more texts than this streams them through ONE tokeniser thread and ONE encoder caller in slabs
# (BgeEncoder.embed_documents_stream): the group commit's first pass leaves with whatever one thread tokenised first - a few
# hundred chunks in a pass that costs as much as a few thousand - and 64 threads take the interpreter lock in turn.
STREAM_MIN_TEXTS = 2048
STREAM_SLAB = 2048

BGE_EMBEDDINGS_MODEL_NAME_OR_PATH = os.environ.get("BGE_EMBEDDINGS_MODEL_PATH", "epam/bge-small-en")

_LAYER_KEYS = [
    "attention.self.query.weight", "attention.self.query.bias",
    "attention.self.key.weight", "attention.self.key.bias",
    "attention.self.value.weight", "attention.self.value.bias",
    "attention.output.dense.weight", "attention.output.dense.bias",
    "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
    "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight", "output.dense.bias",
    "output.LayerNorm.weight", "output.LayerNorm.bias",
]


def _np32(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().float().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


_QC_LOCK = threading.Lock()


class BgeEncoder:
    """Owner of one ``mir_encoder`` handle plus the host-side tokenizer."""

    def __init__(self, handle, layers: int, tokenizer=None, device: int = 0):
        self._h, self.layers, self.tokenizer, self.device = handle, layers, tokenizer, device

    @classmethod
    def from_state_dict(cls, sd: Dict[str, object], tokenizer=None, device: int = 0, prefix: str = "") -> "BgeEncoder":
        """`sd`: BertModel state dict (numpy arrays or torch tensors); `prefix` e.g. "bert." if nested."""
        g = lambda k: _np32(sd[prefix + k])  # noqa: E731
        word, pos, typ = g("embeddings.word_embeddings.weight"), g("embeddings.position_embeddings.weight"), g("embeddings.token_type_embeddings.weight")
        ln_g, ln_b = g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias")
        layers = 0
        while f"{prefix}encoder.layer.{layers}.attention.self.query.weight" in sd:
            layers += 1
        tensors = [g(f"encoder.layer.{i}.{k}") for i in range(layers) for k in _LAYER_KEYS]
        hidden = word.shape[1]
        inter = tensors[10].shape[0] if tensors else 0
        heads = hidden // 32
        arr = (C.c_void_p * len(tensors))(*[t.ctypes.data for t in tensors])
        h = C.c_void_p()
        type0 = np.ascontiguousarray(typ[0])
        nat.check(nat.lib.mir_encoder_create(hidden, layers, heads, inter, word.shape[0], pos.shape[0], nat.ptr(word), nat.ptr(pos),
                                             nat.ptr(type0), nat.ptr(ln_g), nat.ptr(ln_b), arr, device, C.byref(h)))
        return cls(h, layers, tokenizer, device)

    @classmethod
    def from_pretrained_dir(cls, path: str, device: int = 0) -> "BgeEncoder":
        """Load `model.safetensors` + tokenizer files from a local Hugging Face directory.  The tokenizer is the
        native WordPiece one over the directory's vocab.txt (embeddings/wordpiece.py: BertTokenizer's rules, all host
        cores); texts it hands back (code points beyond the BMP) go to the directory's own Hugging Face tokenizer,
        which is only loaded if that ever happens."""
        import json

        from safetensors.numpy import load_file

        from .wordpiece import WordPieceTokenizer

        sd = load_file(os.path.join(path, "model.safetensors"))
        prefix = "bert." if any(k.startswith("bert.") for k in sd) else ""
        lower = True
        cfg = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(cfg):
            lower = bool(json.load(open(cfg)).get("do_lower_case", True))
        hf = []

        def fallback(texts, **kw):
            if not hf:
                from transformers import AutoTokenizer

                hf.append(AutoTokenizer.from_pretrained(path))
            return hf[0](texts, **kw)

        tok = WordPieceTokenizer.from_vocab_file(os.path.join(path, "vocab.txt"), do_lower_case=lower, fallback=fallback)
        return cls.from_state_dict(sd, tok, device, prefix)

    # ---- token-id level (what crosses the C ABI) ----
    def encode_ids(self, sequences: Sequence[Sequence[int]], normalize: bool = True) -> np.ndarray:
        lens = np.asarray([len(s) for s in sequences], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int32) for s in sequences]) if len(sequences) else np.zeros(0, np.int32))
        return self.encode_packed(flat, lens, normalize)

    def encode_packed(self, flat: np.ndarray, lens: np.ndarray, normalize: bool = True) -> np.ndarray:
        """The C ABI's own layout: all token ids back to back (int32) and the sequences' lengths (int32)."""
        flat = np.ascontiguousarray(flat, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if int(lens.sum()) != flat.shape[0]:
            raise ValueError("encode_packed: the lengths do not add up to the number of token ids")
        out = np.zeros((lens.shape[0], EMBEDDING_LENGTH), np.float32)
        nat.check(nat.lib.mir_encoder_encode(self._h, nat.ptr(flat), nat.ptr(lens), lens.shape[0], 1 if normalize else 0, nat.ptr(out)))
        return out

    def encode_ids_to_device(self, sequences: Sequence[Sequence[int]], out_ptr: int, stream: int = 0, normalize: bool = True) -> None:
        lens = np.asarray([len(s) for s in sequences], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int32) for s in sequences]))
        nat.check(nat.lib.mir_encoder_encode_to_device(self._h, nat.ptr(flat), nat.ptr(lens), len(sequences), 1 if normalize else 0, out_ptr, stream or None))

    def debug_hidden(self, sequences: Sequence[Sequence[int]], run_layers: int):
        lens = np.asarray([len(s) for s in sequences], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int32) for s in sequences]))
        padded = int(sum((l + 31) // 32 * 32 for l in lens))
        pooled = np.zeros((len(sequences), EMBEDDING_LENGTH), np.float32)
        hidden = np.zeros((padded, EMBEDDING_LENGTH), np.float32)
        nat.check(nat.lib.mir_encoder_debug_hidden(self._h, nat.ptr(flat), nat.ptr(lens), len(sequences), run_layers, nat.ptr(pooled), nat.ptr(hidden), padded))
        return pooled, hidden

    # ---- text level ----
    def _tokenize(self, texts: Sequence[str]) -> List[List[int]]:
        if self.tokenizer is None:
            raise RuntimeError("this BgeEncoder was built without a tokenizer; use encode_ids or pass one")
        if hasattr(self.tokenizer, "encode_arrays"):  # the native tokenizer: int32 arrays, no Python lists in between
            return self.tokenizer.encode_arrays(list(texts), MAX_TOKENS)
        enc = self.tokenizer(list(texts), add_special_tokens=True, truncation=True, max_length=MAX_TOKENS)
        return enc["input_ids"]

    def _tokenize_packed(self, texts: Sequence[str]):
        """(flat ids, lens) of an outer batch."""
        if self.tokenizer is not None and hasattr(self.tokenizer, "encode_packed"):
            return self.tokenizer.encode_packed(list(texts), MAX_TOKENS)
        seqs = self._tokenize(texts)
        lens = np.asarray([len(s) for s in seqs], dtype=np.int32)
        return (np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs]) if len(seqs) else np.zeros(0, np.int32)), lens

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return [e.tolist() for e in self.embed_documents_numpy(texts)]

    def embed_documents_numpy(self, texts: List[str]) -> List[np.ndarray]:
        """One outer batch of chunk texts (embeddings.py:79-91; the reference's callers hand 128 at a time,
        batched.py:35-53).  A throughput pass of the encoder costs the same whatever its fill (every workgroup
        walks all weights; a full pass is 3072 token tiles ~ 450 chunks of 220 tokens), so concurrent outer
        batches - several documents being indexed, or `build_embeddings` keeping its batches in flight - are
        tokenised in their own threads and share passes (group commit).  An embedding does not depend on what
        else rides in its pass (tests/test_gpu_encoder.py::test_batching_is_invariant)."""
        if not texts:
            return []
        texts = [t.replace("\n", " ") for t in texts]  # HuggingFaceBgeEmbeddings.embed_documents
        return list(self._doc_commit().submit(self._tokenize_packed(texts))[0])

    def embed_documents_stream(self, texts: List[str], on_batches: Optional[Callable[[int], None]] = None) -> List[np.ndarray]:
        """All chunk texts of ONE large document set (``build_embeddings`` with more than STREAM_MIN_TEXTS texts): a tokeniser
        thread runs ahead in slabs of STREAM_SLAB texts (the first one a quarter as long, so the GPU starts early) while this
        thread feeds the encoder slab after slab - the GPU waits for the first slab's tokenisation only.  The same
        embeddings as ``embed_documents_numpy`` batch by batch (test_batching_is_invariant: an embedding does not depend on
        what rides in its pass); `on_batches(n)` reports every n outer batches done, in order."""
        import queue

        texts = [t.replace("\n", " ") for t in texts]
        cuts, at = [], 0
        while at < len(texts):
            step = STREAM_SLAB // 4 if at == 0 else STREAM_SLAB
            cuts.append((at, min(len(texts), at + step)))
            at += step
        q: "queue.Queue" = queue.Queue(maxsize=3)

        def tokenise():
            try:
                for a, b in cuts:
                    q.put((a, b, self._tokenize_packed(texts[a:b])))
            except BaseException as e:  # the consumer re-raises it
                q.put(e)

        t = threading.Thread(target=tokenise, name="bge-tokenise", daemon=True)
        t.start()
        out: List[np.ndarray] = []
        for _ in cuts:
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            a, b, (flat, lens) = item
            out.extend(self.encode_packed(flat, lens))
            if on_batches is not None:
                on_batches((b + EMBEDDINGS_BATCH_SIZE - 1) // EMBEDDINGS_BATCH_SIZE - (a + EMBEDDINGS_BATCH_SIZE - 1) // EMBEDDINGS_BATCH_SIZE)
        t.join()
        return out

    def _doc_commit(self):
        gc = getattr(self, "_dc", None)
        if gc is None:
            with _QC_LOCK:
                gc = getattr(self, "_dc", None)
                if gc is None:
                    from ..retrievers._group_commit import _GroupCommit  # lazy: retrievers imports this module

                    def run(batches):  # list of outer batches, each (flat ids, lens) -> one encode
                        emb = self.encode_packed(np.concatenate([f for f, _ in batches]), np.concatenate([l for _, l in batches]))
                        out, at = [], 0
                        for _, l in batches:
                            out.append(emb[at : at + l.shape[0]])
                            at += l.shape[0]
                        return (out,)

                    gc = self._dc = _GroupCommit(run, max_batch=DOC_BATCHES_PER_PASS)
        return gc

    def embed_query(self, text: str) -> List[float]:
        """One query (embeddings.py:93-96).  Encoding 1 or 16 short queries costs the same ~0.45 ms pass, and the
        reference calls this from concurrent requests: concurrent callers share passes (group commit)."""
        text = text.replace("\n", " ")  # HuggingFaceBgeEmbeddings.embed_query
        ids = self._tokenize([BGE_QUERY_INSTRUCTION_EN + text])[0]
        return self._query_commit().submit(ids)[0].tolist()

    def _query_commit(self):
        gc = getattr(self, "_qc", None)
        if gc is None:
            with _QC_LOCK:
                gc = getattr(self, "_qc", None)
                if gc is None:
                    from ..retrievers._group_commit import _GroupCommit  # lazy: retrievers imports this module

                    gc = self._qc = _GroupCommit(lambda seqs: (self.encode_ids(seqs),), max_batch=256)
        return gc

    def close(self):
        if self._h:
            nat.lib.mir_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_impl: Optional[BgeEncoder] = None


def set_bge_embedding_impl(encoder: BgeEncoder) -> None:
    """Install the process-wide encoder (the reference builds its own at import, embeddings.py:52-66)."""
    global _impl
    _impl = encoder


def bge_embedding_impl() -> BgeEncoder:
    global _impl
    if _impl is None:
        device = detect_device(os.environ.get("BGE_EMBEDDINGS_DEVICE", DeviceType.AUTO))
        if device not in (DeviceType.ROCM, DeviceType.CUDA):
            raise RuntimeError(f"BGE embeddings device {device}: this build computes on a ROCm GPU only")
        if not os.path.isdir(BGE_EMBEDDINGS_MODEL_NAME_OR_PATH):
            raise RuntimeError(
                f"no local model directory at BGE_EMBEDDINGS_MODEL_PATH={BGE_EMBEDDINGS_MODEL_NAME_OR_PATH!r}; "
                "there is no network access to download epam/bge-small-en"
            )
        _impl = BgeEncoder.from_pretrained_dir(BGE_EMBEDDINGS_MODEL_NAME_OR_PATH)
    return _impl


_indexing_pool = None


def _indexing_executor():
    """Threads that tokenise one outer batch each and then wait for their shared encoder pass.  Separate from the
    default executor (where `find` and BM25 run, semantic_retriever.py:54-56) like the reference's
    indexing_embeddings_pool (cpu_pools.py:25-30, 50-53) - but not ONE thread: the GPU pass is shared, the CPU work
    before it (tokenisation) is what the threads parallelise."""
    global _indexing_pool
    if _indexing_pool is None:
        with _QC_LOCK:
            if _indexing_pool is None:
                from concurrent.futures import ThreadPoolExecutor

                _indexing_pool = ThreadPoolExecutor(max_workers=BUILD_IN_FLIGHT, thread_name_prefix="bge-index")
    return _indexing_pool


class AsyncEmbeddings:
    """embeddings.py:72-96.  The sync methods raise upstream too."""

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        raise NotImplementedError()

    def embed_query(self, text: str) -> List[float]:
        raise NotImplementedError()

    async def aembed_documents(self, texts: List[str]) -> List[List[float]]:
        return await asyncio.get_running_loop().run_in_executor(_indexing_executor(), bge_embedding_impl().embed_documents, texts)

    async def aembed_documents_numpy(self, texts: List[str]) -> List[np.ndarray]:
        return await asyncio.get_running_loop().run_in_executor(_indexing_executor(), bge_embedding_impl().embed_documents_numpy, texts)

    async def aembed_query(self, text: str) -> List[float]:
        return await asyncio.get_running_loop().run_in_executor(None, bge_embedding_impl().embed_query, text)


bge_embedding = AsyncEmbeddings()


def _progress(n_batches: int, stageio):
    """The reference's TqdmProgressBar (batched.py:9-28): one line per update, no carriage returns."""
    if stageio is None:
        return None
    from tqdm.std import tqdm as std_tqdm

    class Bar(std_tqdm):
        @staticmethod
        def status_printer(file):
            return file.write

    return Bar(total=n_batches, file=stageio, bar_format="{l_bar}{r_bar}\n", mininterval=10, maxinterval=30, smoothing=0.5, position=0)


async def build_embeddings(texts: Iterable[str], stageio=None) -> List[np.ndarray]:
    """embeddings.py:102-108 + batched.py:35-53: the texts go through `aembed_documents_numpy` in outer batches of
    EMBEDDINGS_BATCH_SIZE, results in order, progress per batch.  The reference awaits one batch at a time because
    each is a heavy CPU job; here a batch is ~0.3 of one GPU pass, so up to BUILD_IN_FLIGHT batches are kept in
    flight and coalesce into full passes inside `BgeEncoder.embed_documents_numpy` - what a caller sees (batch
    size, order, progress lines) is unchanged."""
    texts = list(texts)
    batches = [texts[i : i + EMBEDDINGS_BATCH_SIZE] for i in range(0, len(texts), EMBEDDINGS_BATCH_SIZE)]
    if not batches:
        return []
    impl = bge_embedding_impl()
    if len(texts) > STREAM_MIN_TEXTS and hasattr(impl, "embed_documents_stream") and getattr(impl.tokenizer, "encode_packed", None):
        bar = _progress(len(batches), stageio)
        try:
            return await asyncio.get_running_loop().run_in_executor(
                _indexing_executor(), impl.embed_documents_stream, texts, (bar.update if bar is not None else None))
        finally:
            if bar is not None:
                bar.close()
    gate = asyncio.Semaphore(BUILD_IN_FLIGHT)

    async def one(batch):
        async with gate:
            return await bge_embedding.aembed_documents_numpy(batch)

    tasks = [asyncio.ensure_future(one(b)) for b in batches]
    bar = _progress(len(batches), stageio)
    out: List[np.ndarray] = []
    try:
        for t in tasks:
            out.extend(await t)
            if bar is not None:
                bar.update(1)
    except BaseException:
        for t in tasks:
            t.cancel()
        raise
    finally:
        if bar is not None:
            bar.close()
    return out
