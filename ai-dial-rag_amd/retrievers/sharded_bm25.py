"""Document-sharded BM25 and the sharded hybrid (BASELINE config 4: semantic + BM25 + fusion over 8 GPUs).

Does not exist in the reference (single process).  SURVEY.md 8(e): shard DOCUMENTS, not terms - every rank holds
the postings of a contiguous range of the flattened (doc_record, chunk) order of ``BM25Retriever.from_doc_records``
(aidial_rag/retrievers/bm25_retriever.py:64-79) - while ``idf``, its average and ``avgdl`` stay statistics of the
WHOLE corpus, exchanged once at build time:

    n_t[V]       all-reduce SUM   (document frequency per term)
    first[V]     all-reduce MIN   (global position of each term's first token: rank-bm25 sums the idf average in
                                   dict insertion order, which fixes its float64 rounding)
    tokens, N    all-reduce SUM

after which every rank runs the same host routine (``mir_bm25_idf_from_stats``, the one ``mir_bm25_create`` uses)
and installs the result (``mir_bm25_set_global_stats``: posting weights re-derived on the device for the global
avgdl).  Scores are then bit-identical to the unsharded model.  A query step is the vector path's: per-shard top-k
with the reversed tie-break (ties, the zero tail included, to the HIGHEST global index, bm25_retriever.py:84), one
all-gather of a fused {score, index, count} blob, and ``mir_topk_merge`` with ``descending_scores = 1``.

``ShardedHybrid`` runs both legs on the same stream and fuses on the host with ``mir_rrf_fuse_batch``
(retrieval_chain.py:239-245: weights 1.0, c = 60) - rank fusion is <= 28 items per query.

torch is plumbing: device buffers, the current stream, ``torch.distributed`` ("nccl" = RCCL; "gloo" on CPU in tests).
"""

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from .. import _native as nat
from .sharded_index import DistCollective, _blob_layout

_I64_MAX = np.iinfo(np.int64).max


def idf_from_stats(df: np.ndarray, first_pos: np.ndarray, n_docs: int, epsilon: float = 0.25) -> Tuple[np.ndarray, float]:
    """BM25Okapi._calc_idf from corpus statistics, through the library's own host routine."""
    df = np.ascontiguousarray(df, dtype=np.int64)
    first_pos = np.ascontiguousarray(first_pos, dtype=np.int64)
    idf = np.zeros(len(df), np.float64)
    avg = C.c_double(0.0)
    nat.check(nat.lib.mir_bm25_idf_from_stats(nat.ptr(df), nat.ptr(first_pos), len(df), int(n_docs), float(epsilon), nat.ptr(idf), C.byref(avg)))
    return idf, float(avg.value)


def exchange_global_stats(local, vocab: int, group=None, device="cpu", epsilon: float = 0.25):
    """All-reduce the corpus statistics of `local` (anything with ``corpus_stats()`` / ``set_global_stats()``: a
    ``DeviceBM25`` shard, or the oracle stand-in of the CPU tests) over the process group and install the global
    idf / avgdl.  Returns (idf, avgdl, average_idf, n_docs_global).  Raises the reference's ValueError on every rank
    when the whole corpus has no token (bm25_retriever.py:75-76)."""
    import torch
    import torch.distributed as dist

    df, first, total, n_docs = local.corpus_stats()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    t_df = torch.from_numpy(np.ascontiguousarray(df, dtype=np.int64)).to(device)
    t_cnt = torch.tensor([total, n_docs], dtype=torch.int64, device=device)
    if world > 1:
        every = [torch.zeros_like(t_cnt) for _ in range(world)]
        dist.all_gather(every, t_cnt, group=group)
        token_offset = int(sum(int(e[0].item()) for e in every[:rank]))  # tokens of the shards before this one
    else:
        token_offset = 0
    first = np.asarray(first, dtype=np.int64)
    t_first = torch.from_numpy(np.where(first == _I64_MAX, _I64_MAX, first + token_offset)).to(device)
    if world > 1:
        dist.all_reduce(t_df, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(t_first, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(t_cnt, op=dist.ReduceOp.SUM, group=group)
    total_g, n_g = int(t_cnt[0].item()), int(t_cnt[1].item())
    if total_g == 0:
        raise ValueError("Text index is empty.")
    idf, avg_idf = idf_from_stats(t_df.cpu().numpy(), t_first.cpu().numpy(), n_g, epsilon)
    avgdl = total_g / n_g
    local.set_global_stats(idf, avgdl, avg_idf)
    return idf, avgdl, avg_idf, n_g


def install_combined_stats(shards, epsilon: float = 0.25):
    """Several document shards held by ONE process (in global document order): combine their corpus statistics the
    way `exchange_global_stats` does across ranks - df summed, every term's first token position taken globally,
    tokens and documents summed - and install the global idf / avgdl on each.  -> (idf, avgdl, average_idf, n_docs)."""
    stats = [s.corpus_stats() for s in shards]
    total, n_docs = sum(int(s[2]) for s in stats), sum(int(s[3]) for s in stats)
    if total == 0:
        raise ValueError("Text index is empty.")
    df = np.sum([np.asarray(s[0], np.int64) for s in stats], axis=0)
    offs = np.cumsum([0] + [int(s[2]) for s in stats])
    first = np.min([np.where(np.asarray(s[1]) == _I64_MAX, _I64_MAX, np.asarray(s[1], np.int64) + o) for s, o in zip(stats, offs)], axis=0)
    idf, avg_idf = idf_from_stats(df, first, n_docs, epsilon)
    for s in shards:
        s.set_global_stats(idf, total / n_docs, avg_idf)
    return idf, total / n_docs, avg_idf, n_docs


class ShardedBM25:
    """BM25 top-k over a corpus sharded by document across the ranks of a process group.

    `local_model`: this rank's ``DeviceBM25`` (built with ``doc_offset`` = its first global document), or None when
    `local_search` is given: a callable ``(queries: list of term-id lists, k) -> (idx[b,k] global, score[b,k],
    count[b])`` that stands in for the HIP kernels in the CPU tests."""

    def __init__(self, local_model=None, group=None, device: Optional[str] = None, local_search=None, collective=None):
        import torch

        self.torch = torch
        self.collective = collective if collective is not None else DistCollective(group)
        self.world, self.rank = self.collective.world, self.collective.rank
        if (local_model is None) == (local_search is None):
            raise ValueError("give exactly one of local_model / local_search")
        self.model, self.local_search = local_model, local_search
        self.on_gpu = local_model is not None
        self.device = torch.device(device or (f"cuda:{local_model.device}" if self.on_gpu else "cpu"))
        self._bufs = {}

    @classmethod
    def build(cls, indptr_local, term_ids_local, vocab: int, doc_offset: int, group=None, device_index: int = 0,
              k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25) -> "ShardedBM25":
        """Build this rank's shard on its GPU, exchange the statistics, install the global ones."""
        from .bm25_retriever import DeviceBM25

        # placeholder statistics for the local build (an empty shard is legal; the global corpus is checked below)
        local = DeviceBM25.from_token_ids(indptr_local, term_ids_local, vocab, k1=k1, b=b, epsilon=epsilon,
                                          idf=np.zeros(vocab), avgdl=1.0, device=device_index, doc_offset=doc_offset)
        exchange_global_stats(local, vocab, group, f"cuda:{device_index}", epsilon)
        return cls(local_model=local, group=group)

    def _buffers(self, b: int, k: int):
        key = (b, k)
        if key not in self._bufs:
            t = self.torch
            _, _, size = _blob_layout(b, k)
            ws = self.model.workspace_bytes(b, k) if self.on_gpu else 8
            self._bufs[key] = (
                t.zeros(size // 8, dtype=t.int64, device=self.device),
                t.zeros(self.world * size // 8, dtype=t.int64, device=self.device),
                t.zeros((b, k), dtype=t.float64, device=self.device),
                t.zeros((b, k), dtype=t.int64, device=self.device),
                t.zeros(b, dtype=t.int32, device=self.device),
                t.zeros((ws + 7) // 8, dtype=t.int64, device=self.device),
            )
        return self._bufs[key]

    def search(self, queries, k: int, q_ptr=None):
        """GPU path: `queries` = int32 tensor of all term ids back to back and `q_ptr` = int32 [b + 1] tensor, both on
        the device.  CPU path: a list of term-id lists.  Returns (score[b,k] f64, idx[b,k] i64 global, count[b] i32)
        tensors (asynchronous on the current stream on the GPU path)."""
        t = self.torch
        b = int(q_ptr.numel()) - 1 if q_ptr is not None else len(queries)
        off_idx, off_cnt, size = _blob_layout(b, k)
        blob, gathered, o_score, o_idx, o_cnt, ws = self._buffers(b, k)
        if self.on_gpu and self.world == 1:  # one shard: its top-k is the result
            self.model.search_device(queries.data_ptr(), q_ptr.data_ptr(), b, k, o_idx.data_ptr(), o_score.data_ptr(), o_cnt.data_ptr(),
                                     ws.data_ptr(), t.cuda.current_stream(self.device).cuda_stream)
            return o_score, o_idx, o_cnt
        if self.on_gpu:
            stream = t.cuda.current_stream(self.device).cuda_stream
            base = blob.data_ptr()
            self.model.search_device(queries.data_ptr(), q_ptr.data_ptr(), b, k, base + off_idx, base, base + off_cnt,
                                     ws.data_ptr(), stream)
        else:
            i_, s_, c_ = self.local_search(queries, k)
            raw = blob.numpy().view(np.uint8)
            raw[:off_idx].view(np.float64)[:] = np.asarray(s_, np.float64).reshape(-1)
            raw[off_idx:off_cnt].view(np.int64)[:] = np.asarray(i_, np.int64).reshape(-1)
            raw[off_cnt : off_cnt + 4 * b].view(np.int32)[:] = np.asarray(c_, np.int32)
        if self.world == 1:
            src = blob
        else:
            self.collective.all_gather_into_tensor(gathered, blob)
            src = gathered
        if self.on_gpu:
            base = src.data_ptr()
            nat.check(nat.lib.mir_topk_merge_device(base, base + off_idx, base + off_cnt, self.world, size, b, k, 1,
                                                    o_score.data_ptr(), o_idx.data_ptr(), o_cnt.data_ptr(), self.model.device,
                                                    t.cuda.current_stream(self.device).cuda_stream))
        else:
            raw = src.numpy().view(np.uint8)
            base = raw.ctypes.data
            nat.check(nat.lib.mir_topk_merge_host(base, base + off_idx, base + off_cnt, self.world, size, b, k, 1,
                                                  o_score.data_ptr(), o_idx.data_ptr(), o_cnt.data_ptr()))
        return o_score, o_idx, o_cnt


def fuse_batch(lists: Sequence[Tuple[np.ndarray, np.ndarray]], weights: Sequence[float], c: int = 60):
    """Weighted reciprocal-rank fusion of `len(lists)` retrievers' results for b queries at once.

    lists[l] = (ids[b, k_l] int64, count[b]): retriever l's ranked ids per query (an id = a chunk's global position;
    the reference keys by "{doc_id}_{chunk_id}", index_record.py:33-34 - here the pair is (id, 0)).
    Returns (fused ids [b, cap] int64, scores [b, cap], count [b])."""
    b = len(lists[0][1])
    nl = len(lists)
    lp = np.zeros((b, nl + 1), np.int32)
    for l, (_, cnt) in enumerate(lists):
        lp[:, l + 1] = lp[:, l] + np.asarray(cnt, np.int32)
    cap = int(sum(ids.shape[1] for ids, _ in lists))
    keys = np.zeros((b, cap, 2), np.int64)
    if all(int(np.min(cnt, initial=ids.shape[1])) == ids.shape[1] for ids, cnt in lists):
        # every list full (the usual case): the lists side by side, no per-element placement
        at = 0
        for ids, _ in lists:
            keys[:, at : at + ids.shape[1], 0] = ids
            at += ids.shape[1]
    else:
        for l, (ids, cnt) in enumerate(lists):
            ids = np.asarray(ids)
            kl = ids.shape[1]
            col = np.arange(kl)[None, :]
            ok = col < np.asarray(cnt)[:, None]
            dst = lp[:, l][:, None] + col
            qq, cc = np.nonzero(ok)
            keys[qq, dst[qq, cc], 0] = ids[qq, cc]
    key_base = (np.arange(b, dtype=np.int64) * cap)
    out_keys = np.zeros((b, cap, 2), np.int64)
    out_scores = np.zeros((b, cap), np.float64)
    out_cnt = np.zeros(b, np.int32)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    nat.check(nat.lib.mir_rrf_fuse_batch(nat.ptr(keys), nat.ptr(key_base), nat.ptr(np.ascontiguousarray(lp)), nat.ptr(w), nl, c, b, cap,
                                         nat.ptr(out_keys), nat.ptr(out_scores), nat.ptr(out_cnt)))
    return out_keys[:, :, 0], out_scores, out_cnt


class ShardedHybrid:
    """Semantic + BM25 + fusion over a sharded corpus: BASELINE config 4's step.  `vector` is a ``ShardedSearcher``,
    `keywords` a ``ShardedBM25`` over the same chunks in the same global order; both legs are enqueued on the current
    stream, their merged top-k come to the host in one synchronisation and are fused there (every rank holds the
    merged lists after its local merge; fusion is replicated, not communicated)."""

    def __init__(self, vector, keywords, k: int = 7, weights: Sequence[float] = (1.0, 1.0), c: int = 60):
        self.vector, self.keywords, self.k, self.weights, self.c = vector, keywords, k, tuple(weights), c

    def search(self, query_vectors, metric, query_terms, q_ptr=None):
        """-> (fused ids [b, 2k], scores [b, 2k], count [b]) numpy arrays, plus the two legs' id lists."""
        _, v_rows, v_cnt, _ = self.vector.search(query_vectors, self.k, metric)
        _, t_idx, t_cnt = self.keywords.search(query_terms, self.k, q_ptr)
        if self.vector.on_gpu:
            # both legs' lists to the host in ONE copy: [vector rows | BM25 rows | counts (int32 pairs)] as int64
            tt = self.vector.torch
            b, k = int(v_rows.shape[0]), self.k
            key = (b, k)
            if getattr(self, "_stage_key", None) != key:
                self._stage = tt.empty(b * (2 * k + 1), dtype=tt.int64, device=self.vector.device)
                self._stage_host = tt.empty(b * (2 * k + 1), dtype=tt.int64, pin_memory=True)
                self._stage_key = key
            st = self._stage
            st[: b * k].copy_(v_rows.reshape(-1))
            st[b * k : 2 * b * k].copy_(t_idx.reshape(-1))
            cnts = st[2 * b * k :].view(tt.int32).view(b, 2)
            cnts[:, 0].copy_(v_cnt)
            cnts[:, 1].copy_(t_cnt)
            self._stage_host.copy_(st, non_blocking=True)
            tt.cuda.current_stream(self.vector.device).synchronize()
            h = self._stage_host.numpy()
            hc = h[2 * b * k :].view(np.int32).reshape(b, 2)
            v = (h[: b * k].reshape(b, k).copy(), hc[:, 0].copy())
            t = (h[b * k : 2 * b * k].reshape(b, k).copy(), hc[:, 1].copy())
        else:
            v = (v_rows.cpu().numpy(), v_cnt.cpu().numpy())
            t = (t_idx.cpu().numpy(), t_cnt.cpu().numpy())
        ids, scores, cnt = fuse_batch([v, t], self.weights, self.c)
        return ids, scores, cnt, v, t
