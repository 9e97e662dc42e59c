"""Where a multi-rank hybrid step spends its time, two ranks on ONE GPU over gloo (a rehearsal: RCCL is what it leaves out):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 tools/two_rank_hybrid_timing.py [docs_per_rank]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bench
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher
from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25, ShardedHybrid

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
n_loc, d, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000, 384, 128, 7
g = torch.Generator(device=dev); g.manual_seed(555 + rank)
rows = torch.randn((n_loc, d), generator=g, dtype=torch.float32, device=dev)
rows /= rows.norm(dim=1, keepdim=True)
index = DeviceIndex.from_device_ptr(rows.data_ptr(), n_loc, d, 0, row_offset=rank * n_loc, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
indptr, toks = bench.gen_bm25_corpus(np, torch, dev, n_loc, 9000 + rank)
kw = ShardedBM25.build(indptr, toks, bench.BM25_VOCAB, doc_offset=rank * n_loc, device_index=0)
se = ShardedSearcher(local_index=index)
hy = ShardedHybrid(se, kw, k=k)
g.manual_seed(4321)
q = torch.randn((B, d), generator=g, dtype=torch.float32, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).double().contiguous()
qs = bench.bm25_queries(np, B, 778)
flat = torch.tensor(np.concatenate([np.asarray(x, np.int32) for x in qs]), dtype=torch.int32, device=dev)
ptr = torch.tensor(np.concatenate(([0], np.cumsum([len(x) for x in qs]))), dtype=torch.int32, device=dev)


def timed(name, fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); dist.barrier()
    if rank == 0: print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / reps:9.3f} ms per call", flush=True)


blob, gathered, *_ = kw._buffers(B, k)
o_row = torch.zeros((B, k), dtype=torch.int64, device=dev); o_dist = torch.zeros((B, k), dtype=torch.float64, device=dev)
o_cnt = torch.zeros(B, dtype=torch.int32, device=dev); o_flag = torch.zeros(B, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
timed("local vector search only, no sync", lambda: index.search_device(q.data_ptr(), B, k, "sqeuclidean_dist", out_row_ptr=o_row.data_ptr(), out_dist_ptr=o_dist.data_ptr(),
                                                                          out_count_ptr=o_cnt.data_ptr(), out_flags_ptr=o_flag.data_ptr(), stream=stream))
timed("local vector search + stream sync", lambda: (index.search_device(q.data_ptr(), B, k, "sqeuclidean_dist", out_row_ptr=o_row.data_ptr(), out_dist_ptr=o_dist.data_ptr(),
                                                                          out_count_ptr=o_cnt.data_ptr(), out_flags_ptr=o_flag.data_ptr(), stream=stream), torch.cuda.synchronize()))
timed("vector leg (search + gather + merge)", lambda: se.search(q, k, "sqeuclidean_dist"))
timed("BM25 leg (search + gather + merge)", lambda: kw.search(flat, k, ptr))
timed("all_gather_into_tensor alone", lambda: dist.all_gather_into_tensor(gathered, blob))
timed("hybrid step", lambda: hy.search(q, "sqeuclidean_dist", flat, ptr))
