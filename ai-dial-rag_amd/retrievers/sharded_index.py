"""Row-sharded vector search: one process per GPU, one exchange step.

Does not exist in the reference (it is single-process); it is the reference's
own two-level structure - per-document partial top-k, then a second stable
argsort over the concatenation (aidial_rag/retrievers/embeddings_index.py:
62-81) - applied across GPUs: every rank owns a contiguous range of the
flattened (doc, row) order, finds its exact partial top-k, and one RCCL
all-gather of ``B x k x (f64 dist, i64 row) + B x i32 count`` per rank is
followed by a local k-way merge on (distance, global row), which keeps the
reference's tie-break global.  The all-gather is latency-bound (a few KB), so
it is a single fused blob per rank, not three collectives.

torch is plumbing here: device buffers, the current stream and
``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" on CPU in tests).
"""

from typing import Callable, Optional, Tuple

import numpy as np

from .. import _native as nat
from .embeddings_metrics import Metric


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of `rank` in the flattened order."""
    return (n * rank) // world, (n * (rank + 1)) // world


class DistCollective:
    """The exchange step over ``torch.distributed`` ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def all_gather_into_tensor(self, gathered, blob):
        self.dist.all_gather_into_tensor(gathered, blob, group=self.group)


class LoopbackCollective:
    """`world` shards of ONE process (one thread per shard, all on the same device and stream) exchanging their blobs
    by device-to-device copies: the layout ``all_gather_into_tensor`` produces - rank s's blob at offset s * size of
    every rank's gathered tensor - without a process group.  It exists so that the `world > 1` branch of the sharded
    searchers (per-shard search into a blob, gathered blobs with a non-zero shard stride, the library merge) runs on
    the real kernels of a single GPU (tests/test_gpu_sharded_c4.py); `for_rank(r)` is what shard r's searcher gets."""

    def __init__(self, world: int):
        import threading

        self.world = world
        self._barrier = threading.Barrier(world)
        self._blobs = [None] * world

    def for_rank(self, rank: int):
        return _LoopbackRank(self, rank)


class _LoopbackRank:
    def __init__(self, bus: LoopbackCollective, rank: int):
        self.bus, self.rank, self.world = bus, rank, bus.world

    def all_gather_into_tensor(self, gathered, blob):
        bus = self.bus
        bus._blobs[self.rank] = blob
        bus._barrier.wait()  # every shard's search is enqueued (same stream: the copies below run behind all of them)
        n = blob.numel()
        for s in range(self.world):
            gathered[s * n : (s + 1) * n].copy_(bus._blobs[s])
        bus._barrier.wait()  # nobody reuses its blob before every rank has enqueued its copies


def _blob_layout(b: int, k: int):
    """Byte offsets of {dist f64[b][k], row i64[b][k], count i32[b]} and the padded blob size."""
    off_row = b * k * 8
    off_cnt = 2 * b * k * 8
    size = off_cnt + ((b * 4 + 7) // 8) * 8
    return off_row, off_cnt, size


class ShardedSearcher:
    """Search over an index sharded by row across the ranks of a process group.

    `local_index` is this rank's ``DeviceIndex`` (created with
    ``row_offset = shard_bounds(...)[0]``), or None when `local_search` is
    given: a callable ``(queries f64[b,d], k, metric) -> (dist[b,k], rows[b,k],
    count[b])`` used by the CPU tests to stand in for the HIP kernels.
    """

    def __init__(self, local_index=None, group=None, device: Optional[str] = None,
                 local_search: Optional[Callable] = None, collective=None):
        import torch

        self.torch = torch
        self.collective = collective if collective is not None else DistCollective(group)
        self.world, self.rank = self.collective.world, self.collective.rank
        self.index = local_index
        self.local_search = local_search
        if (local_index is None) == (local_search is None):
            raise ValueError("give exactly one of local_index / local_search")
        self.on_gpu = local_index is not None
        self.device = torch.device(device or (f"cuda:{local_index.device}" if self.on_gpu else "cpu"))
        self._bufs = {}

    def _buffers(self, b: int, k: int):
        key = (b, k)
        if key not in self._bufs:
            t = self.torch
            _, _, size = _blob_layout(b, k)
            self._bufs[key] = (
                t.zeros(size // 8, dtype=t.int64, device=self.device),                # this rank's blob
                t.zeros(self.world * size // 8, dtype=t.int64, device=self.device),   # gathered blobs
                t.zeros((b, k), dtype=t.float64, device=self.device),
                t.zeros((b, k), dtype=t.int64, device=self.device),
                t.zeros(b, dtype=t.int32, device=self.device),
                t.zeros(b, dtype=t.int32, device=self.device),
            )
        return self._bufs[key]

    def search(self, queries, k: int, metric="sqeuclidean_dist", out_flags=None):
        """queries: float64 [b, d] torch tensor on `self.device` (GPU path) or array-like (CPU path).
        Returns (dist[b,k] f64, rows[b,k] i64 global, count[b] i32, flags[b] i32) tensors; asynchronous on
        the current stream on the GPU path.  `out_flags` (GPU path): an int32 [b] tensor to receive this call's
        flags instead of the searcher's own buffer, which the next call overwrites.  The flags are THIS rank's and
        informational (MIR_FLAG_EXACT_PASS: the shard answered the query by its exact pass): every shard's top-k is the
        exact one either way (vec_index.hip), so the merged result does not depend on another rank's flags and the
        all-gather does not carry them."""
        t = self.torch
        metric = Metric(metric).value
        b = int(queries.shape[0])
        off_row, off_cnt, size = _blob_layout(b, k)
        blob, gathered, o_dist, o_row, o_cnt, o_flags = self._buffers(b, k)
        if out_flags is not None and self.on_gpu:
            if out_flags.dtype != t.int32 or out_flags.numel() != b or not out_flags.is_contiguous():
                raise ValueError("out_flags must be a contiguous int32 tensor of b elements")
            o_flags = out_flags
        if self.on_gpu and self.world == 1:
            # one shard: its exact top-k IS the result - no blob, no merge dispatch (a dependent ~8 us launch per step)
            self.index.search_device(queries.data_ptr(), b, k, metric, out_row_ptr=o_row.data_ptr(), out_dist_ptr=o_dist.data_ptr(),
                                     out_count_ptr=o_cnt.data_ptr(), out_flags_ptr=o_flags.data_ptr(),
                                     stream=t.cuda.current_stream(self.device).cuda_stream)
            return o_dist, o_row, o_cnt, o_flags
        if self.on_gpu:
            stream = t.cuda.current_stream(self.device).cuda_stream
            base = blob.data_ptr()
            self.index.search_device(queries.data_ptr(), b, k, metric, out_row_ptr=base + off_row, out_dist_ptr=base,
                                     out_count_ptr=base + off_cnt, out_flags_ptr=o_flags.data_ptr(), stream=stream)
        else:
            d_, r_, c_ = self.local_search(np.asarray(queries, dtype=np.float64), k, metric)
            raw = blob.numpy().view(np.uint8)
            raw[:off_row].view(np.float64)[:] = np.asarray(d_, np.float64).reshape(-1)
            raw[off_row:off_cnt].view(np.int64)[:] = np.asarray(r_, np.int64).reshape(-1)
            raw[off_cnt : off_cnt + 4 * b].view(np.int32)[:] = np.asarray(c_, np.int32)
        if self.world == 1:
            src = blob
        else:
            self.collective.all_gather_into_tensor(gathered, blob)
            src = gathered
        if self.on_gpu:
            base = src.data_ptr()
            nat.check(nat.lib.mir_topk_merge_device(base, base + off_row, base + off_cnt, self.world, size, b, k, 0,
                                                    o_dist.data_ptr(), o_row.data_ptr(), o_cnt.data_ptr(),
                                                    self.index.device, t.cuda.current_stream(self.device).cuda_stream))
        else:
            raw = src.numpy().view(np.uint8)
            base = raw.ctypes.data
            nat.check(nat.lib.mir_topk_merge_host(base, base + off_row, base + off_cnt, self.world, size, b, k, 0,
                                                  o_dist.data_ptr(), o_row.data_ptr(), o_cnt.data_ptr()))
        return o_dist, o_row, o_cnt, o_flags
