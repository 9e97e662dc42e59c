#!/usr/bin/env python3
"""Headline benchmark: retrieval QPS over a 10M x 384 float32 index (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of B synthetic queries (256 by
default: one launch group of the sieve): the filter over this rank's row shard,
the float64 reference formula for the rows that can be among the first k, their
reference order, and (N > 1) one RCCL all-gather of per-shard partial top-k plus
a local merge.  The index
and the queries are resident in HBM before the timed region.  The total index
is fixed at --rows as N grows ("strong" scaling: the shard is rows / N).
Before the W warm-up steps the same step runs untimed until 12 launches have
been made in all (`preconditioning_steps` in the output): from idle the GPU
takes ~10 launches to reach its steady clock, and a short --warmup would
otherwise time that ramp.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the scan)
from HIP events recorded on its launch stream inside the timed region;
`cpu_baseline` times the CPU oracle (numpy restatement of the reference path,
float64 query exactly as the live path) on a bounded sample on the host cores.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS_BF16 = 2500.0  # dense bf16 / f16 MFMA peak (MI355X_MICROARCH.md; never the 2:1-sparsity figure)
MFMA_PEAK_TOPS_I8 = 5000.0      # dense int8 MFMA peak: v_mfma_i32_16x16x64_i8 runs at twice the bf16 rate (MI355X_MICROARCH.md, matrix cores)


def sieve_image_bytes_per_element(int8):
    """What the sieve's first stage streams per row element: the int8 image (csrc/vec_kernels_i8.h) or the bf16 hi blocks."""
    return 1 if int8 else 2
MAIN_LEG_CHECK_QUERIES = 2
CHUNK_ROWS = 500_000   # corpus is generated in fixed global chunks so every N sees the same rows
PRECONDITION_STEPS = 12  # untimed launches before timing starts (W of them are the warm-up steps), see main()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=256,
                    help="queries per step (B).  One pass of the shard serves up to 256 queries (two 16-query tiles per wave): the "
                         "stream is the same as for 128, the matrix work doubles, so 256 gives the most QPS (~137k; 128: ~95k; 64: ~52k); "
                         "the sweep reports the others")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="sqeuclidean_dist")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-queries", type=int, default=16)
    ap.add_argument("--no-sweep", dest="sweep", action="store_false", help="skip the B = 1 / 32 / 64 side measurements")
    ap.add_argument("--c5-rows", type=int, default=6_250_000,
                    help="rows PER GPU of the float16 d=1024 leg (BASELINE config 5: 50M over 8 GPUs); 0 = skip")
    ap.add_argument("--c2-chunks", type=int, default=0,
                    help="BASELINE config C2 at its own size: encode this many chunks (1000000) into one block, index and search it; 0 = skip")
    ap.add_argument("--encode-chunks", type=int, default=8192,
                    help="chunks per GPU for the index-build (encoder) leg of the metric; 0 = skip")
    ap.add_argument("--bm25-docs", type=int, default=1_000_000,
                    help="documents of the BM25 leg (BASELINE config 3: 1M chunks, 50k-term vocabulary), single GPU only; 0 = skip")
    ap.add_argument("--hybrid-docs", type=int, default=1_250_000,
                    help="chunks PER GPU of the hybrid leg (BASELINE config 4: 10M chunks over 8 GPUs = 1.25M each: vector "
                         "rows + BM25 documents + fusion); 0 = skip")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the timed steps are issued on, round-robin (independent batches in flight); 1 = strictly one after the "
                         "other; 0 (default) = 1 for shards of >= 4M rows per GPU (a step is >= 1 ms of streaming: nothing to hide), 3 below")
    ap.add_argument("--no-variants", dest="variants", action="store_false",
                    help="skip the clustered-corpus and near-duplicate-corpus legs of the headline search (single GPU only)")
    ap.add_argument("--no-cpu-legs", dest="cpu_legs", action="store_false",
                    help="skip the BM25 and encoder CPU baselines (the vector one is governed by --cpu-rows)")
    return ap.parse_args()


def gen_rows(torch, device, lo, hi, dim):
    """Unit-norm float32 rows [lo, hi) of the global synthetic corpus, generated on the GPU."""
    out = torch.empty((hi - lo, dim), dtype=torch.float32, device=device)
    c0, c1 = lo // CHUNK_ROWS, (hi - 1) // CHUNK_ROWS
    for c in range(c0, c1 + 1):
        g = torch.Generator(device=device)
        g.manual_seed(1234 + c)
        x = torch.randn((CHUNK_ROWS, dim), generator=g, dtype=torch.float32, device=device)
        x /= x.norm(dim=1, keepdim=True)
        a, b = max(lo, c * CHUNK_ROWS), min(hi, (c + 1) * CHUNK_ROWS)
        out[a - lo : b - lo] = x[a - c * CHUNK_ROWS : b - c * CHUNK_ROWS]
        del x
    return out


def random_bge_small_state_dict(np, seed=0):
    """Random-init weights of the bge-small-en architecture (there is no network for the real checkpoint):
    BERT, hidden 384, 12 layers, 12 heads, FFN 1536, vocab 30522 - Hugging Face parameter names."""
    rng = np.random.default_rng(seed)
    w = lambda *shape: (rng.standard_normal(shape) * 0.02).astype(np.float32)  # noqa: E731
    sd = {
        "embeddings.word_embeddings.weight": w(30522, 384),
        "embeddings.position_embeddings.weight": w(512, 384),
        "embeddings.token_type_embeddings.weight": w(2, 384),
        "embeddings.LayerNorm.weight": np.ones(384, np.float32),
        "embeddings.LayerNorm.bias": np.zeros(384, np.float32),
    }
    for i in range(12):
        p = f"encoder.layer.{i}."
        for name, shape in (("attention.self.query", (384, 384)), ("attention.self.key", (384, 384)),
                            ("attention.self.value", (384, 384)), ("attention.output.dense", (384, 384)),
                            ("intermediate.dense", (1536, 384)), ("output.dense", (384, 1536))):
            sd[p + name + ".weight"] = w(*shape)
            sd[p + name + ".bias"] = np.zeros(shape[0], np.float32)
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + name + ".weight"] = np.ones(384, np.float32)
            sd[p + name + ".bias"] = np.zeros(384, np.float32)
    return sd


def encoder_leg(np, torch, dist, args, world, rank, local_rank, barrier):
    """Index-build half of the metric: chunks/s of the bge-small-en forward (CLS + L2 normalise), every rank
    encoding its own `--encode-chunks` synthetic chunks (token ids, length ~N(220, 60) clipped to [8, 512],
    SURVEY.md 8(d)) straight into HBM.  Pure data parallel: no communication."""
    from aidial_rag_amd.embeddings.embeddings import BgeEncoder

    enc = BgeEncoder.from_state_dict(random_bge_small_state_dict(np), device=local_rank)
    rng = np.random.default_rng(99 + rank)
    n = args.encode_chunks
    lens = np.clip(np.round(rng.normal(220, 60, n)), 8, 512).astype(np.int64)
    seqs = [rng.integers(999, 30522, L).astype(np.int32) for L in lens]
    out = torch.empty((n, 384), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream().cuda_stream
    # warm-up = one call of the same size: the first full-size call also allocates the encoder's workspaces and pinned
    # staging (tens of ms; `through_build_embeddings.chunks_per_s_each_run` shows that first call separately), which an
    # index build that encodes document after document pays once
    enc.encode_ids_to_device(seqs, out.data_ptr(), stream)
    barrier()
    t0 = time.perf_counter()
    enc.encode_ids_to_device(seqs, out.data_ptr(), stream)
    barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    tokens = int(lens.sum())
    flops = float(sum(12 * (L * (2 * 384 * 1152 + 2 * 384 * 384 + 4 * 384 * 1536) + 4 * L * L * 384) for L in lens))
    norms_ok = bool(torch.allclose(out.norm(dim=1), torch.ones(n, device=out.device), atol=1e-4))
    enc.close()
    return {
        "index_build_chunks_per_s": round(world * n / dt, 1),
        "chunks_per_gpu": n,
        "mean_tokens_per_chunk": round(tokens / n, 1),
        "tflops_per_gpu_real_tokens": round(flops / dt / 1e12, 1),
        "mfma_peak_tflops_f16_dense": 2500.0,
        "frac_of_mfma_peak": round(flops / dt / 1e12 / 2500.0, 4),
        "weights": "random init, bge-small-en shape",
        "unit_norm_outputs": norms_ok,
        "warmup": "one untimed call of the same size (workspaces allocated), then one timed call",
    }


def c2_leg(np, torch, args, local_rank, DeviceIndex, ShardedSearcher):
    """BASELINE config C2 at its own size (behind --c2-chunks, e.g. 1000000): encode that many synthetic chunks (token ids,
    length ~N(220, 60)) straight into ONE float32 block in HBM - slabs of 65 536 chunks, only the encoder calls timed - then
    build the index over the block where it lies and search it: chunks/s quoted on a BASELINE size, and the two halves of
    the metric meeting in one place."""
    from aidial_rag_amd.embeddings.embeddings import BgeEncoder

    device = torch.device("cuda", local_rank)
    n, slab = args.c2_chunks, 65536
    enc = BgeEncoder.from_state_dict(random_bge_small_state_dict(np), device=local_rank)
    rng = np.random.default_rng(1999)
    out = torch.empty((n, 384), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    warm = [rng.integers(999, 30522, 220).astype(np.int32) for _ in range(4096)]
    enc.encode_ids_to_device(warm, out.data_ptr(), stream)
    torch.cuda.synchronize()
    dt, tokens, flops = 0.0, 0, 0.0
    for s0 in range(0, n, slab):
        m = min(slab, n - s0)
        lens = np.clip(np.round(rng.normal(220, 60, m)), 8, 512).astype(np.int64)
        seqs = [rng.integers(999, 30522, L).astype(np.int32) for L in lens]  # (untimed: synthetic input generation)
        t0 = time.perf_counter()
        enc.encode_ids_to_device(seqs, out.data_ptr() + s0 * 384 * 4, stream)
        torch.cuda.synchronize()
        dt += time.perf_counter() - t0
        tokens += int(lens.sum())
        flops += float(sum(12 * (L * (2 * 384 * 1152 + 2 * 384 * 384 + 4 * 384 * 1536) + 4 * L * L * 384) for L in lens))
    enc.close()
    t0 = time.perf_counter()
    index = DeviceIndex.from_device_ptr(out.data_ptr(), n, 384, local_rank, stream=stream)
    torch.cuda.synchronize()
    t_index = time.perf_counter() - t0
    B, k = 256, args.k
    g = torch.Generator(device=device)
    g.manual_seed(7)
    pick = torch.randint(0, n, (B,), generator=g, device=device)
    q = out[pick].double()
    q[1::2] += 0.05 * torch.randn((B // 2, 384), generator=g, dtype=torch.float64, device=device)  # every other query: near its row
    q = q.contiguous()
    searcher = ShardedSearcher(local_index=index)
    for _ in range(5):
        res = searcher.search(q, k, args.metric)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        res = searcher.search(q, k, args.metric)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rows = res[1].cpu().numpy()
    found = float((rows[:, :2] == pick.cpu().numpy()[:, None]).any(axis=1).mean())  # (random-weight embeddings crowd together: the row itself or its twin)
    index.close()
    return {"workload": f"{n} synthetic chunks (mean {tokens / n:.1f} tokens) -> float32 {n} x 384 in HBM -> index -> {B}-query search, k={k}",
            "index_build_chunks_per_s": round(n / dt, 1), "encode_s": round(dt, 2),
            "frac_of_mfma_peak": round(flops / dt / 1e12 / 2500.0, 4), "index_from_rows_s": round(t_index, 3),
            "search_qps": round(B * 10 / el, 1), "search_ms_per_step": round(1e3 * el / 10, 4),
            "queries_finding_their_source_row_in_top2": found}


def c5_leg(np, torch, dist, args, world, rank, local_rank, barrier, DeviceIndex, ShardedSearcher):
    """BASELINE config 5: d = 1024 float16 vectors, not normalised, sqeuclidean;
    rows_per_gpu fixed (50M over 8 GPUs = 6.25M each), so this leg scales weakly with --gpus.
    The index keeps the rows in float16 (2 B/element scanned, 128 queries per pass: vec_kernels_h16.h)."""
    device = torch.device("cuda", local_rank)
    n_loc, d, B, k = args.c5_rows, 1024, 128, args.k
    g = torch.Generator(device=device)
    shard = torch.empty((n_loc, d), dtype=torch.float16, device=device)
    for c in range(0, n_loc, 250_000):
        g.manual_seed(2024 + rank * 1000 + c // 250_000)
        m = min(250_000, n_loc - c)
        shard[c : c + m] = torch.randn((m, d), generator=g, dtype=torch.float32, device=device).half()
    torch.cuda.synchronize()
    index = DeviceIndex.from_device_ptr(shard.data_ptr(), n_loc, d, local_rank, row_offset=rank * n_loc,
                                        stream=torch.cuda.current_stream().cuda_stream, float16=True)
    torch.cuda.synchronize()
    hbm = index.hbm_bytes()
    del shard
    torch.cuda.empty_cache()
    g.manual_seed(4322)
    queries = torch.randn((B, d), generator=g, dtype=torch.float32, device=device).double().contiguous()
    searcher = ShardedSearcher(local_index=index)
    index.profile(True)
    for _ in range(PRECONDITION_STEPS):
        out = searcher.search(queries, k, "sqeuclidean_dist")
    barrier()
    index.profile_read(reset=True)
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        out = searcher.search(queries, k, "sqeuclidean_dist")
    barrier()
    elapsed = time.perf_counter() - t0
    index.profile(False)
    launches, scan_ms = index.profile_read(reset=True)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    flags = int(out[3].sum().item())
    index.close()
    bytes_pass = n_loc * d * 2 + 4 * n_loc + 128 * d * 4 + 128 * k * 12  # SURVEY 8(d), s = 2 (fp16), 128 queries per pass
    avg_ms = scan_ms / max(launches, 1)
    return {
        "workload": f"{n_loc} x {d} float16 rows per GPU (not normalised), sqeuclidean_dist, k={k}",
        "scaling": "weak",
        "queries_per_step": B,
        "queries_per_pass": 128,
        "ms_per_step": round(1e3 * elapsed / steps, 4),
        "qps": round(B * steps / elapsed, 1),
        "index_hbm_bytes_per_gpu": hbm,
        "roofline": {"bound": "hbm", "kernel": "sieve_h16_kernel: two filter launches over the float16 image with sieve_scatter_kernel + sieve_select_kernel between them (round 2: scan_topk_h16_kernel with candidate lists; MIR_NO_SIEVE16=1 selects it)", "bytes_per_launch": bytes_pass,
                     "avg_launch_ms": round(avg_ms, 4), "achieved": round(bytes_pass / (avg_ms * 1e-3) / 1e9, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_pass / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "launches": launches},
        "exact_pass_queries": flags // 2,
    }


def gen_clustered_rows(torch, device, n, dim, centres, seed, a=0.9486833, b=0.3162278):
    """Rows drawn from a mixture: unit centres, row = normalise(a * centre + b * unit noise) - cosine ~0.9 between two
    rows of a cluster.  What an index of real embeddings looks like more than isotropic noise does."""
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for c0 in range(0, n, CHUNK_ROWS):
        g.manual_seed(seed + c0 // CHUNK_ROWS)
        m = min(CHUNK_ROWS, n - c0)
        cid = torch.randint(0, centres.shape[0], (m,), generator=g, device=device)
        x = torch.randn((m, dim), generator=g, dtype=torch.float32, device=device)
        x /= x.norm(dim=1, keepdim=True)
        x = a * centres[cid] + b * x
        out[c0 : c0 + m] = x / x.norm(dim=1, keepdim=True)
        del x, cid
    return out


def gen_near_duplicate_rows(torch, device, n, dim, seed, frac=0.10, group=32, eps=1e-7):
    """Isotropic unit rows of which `frac` sit in groups of `group` near-identical rows (boiler-plate pages, repeated
    chunks): a group's rows differ from its first by `eps`-sized noise, i.e. their distances to any query agree to
    ~1e-7 - inside every float32 filter's error band.  Returns (rows, first row of every group)."""
    rows = gen_rows(torch, device, 0, n, dim)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_groups = int(n * frac) // group
    starts = torch.randperm(n // group, generator=g, device=device)[:n_groups] * group
    for c0 in range(0, n_groups, 4096):
        st = starts[c0 : c0 + 4096]
        base = rows[st]
        idx = (st[:, None] + torch.arange(group, device=device)[None, :]).reshape(-1)
        noise = torch.randn((len(idx), dim), generator=g, dtype=torch.float32, device=device) * eps
        x = base.repeat_interleave(group, dim=0) + noise
        x[::group] = base
        rows[idx] = x / x.norm(dim=1, keepdim=True)
    return rows, starts


def variant_leg(np, torch, args, local_rank, DeviceIndex, ShardedSearcher, kind):
    """The headline search (same N x d, k, B, metric) on a corpus that is NOT isotropic noise; single GPU.
      clustered       4096 centres, intra-cluster cosine ~0.9, queries near centres: how much of the scan's speed
                      comes from thresholds that isotropic data makes easy
      near_duplicate  10 % of the rows in groups of 32 near-identical rows, half of the queries aimed at such a
                      group: the k-th neighbour falls inside a group, which no float32 filter can order - the sieve's
                      select evaluates the group in float64 (round 2's list scan sent those queries to the exact pass)
    Reports QPS, the roofline of the scan bracket, the share of queries answered by the exact pass, and - on the
    host, with the CPU oracle over ALL rows - identity of the ids for two queries."""
    from oracle import embeddings_index as oracle_index

    device = torch.device("cuda", local_rank)
    n, d, B, k = args.rows, args.dim, args.batch, args.k
    g = torch.Generator(device=device)
    g.manual_seed(31337)
    pool = 16 * B
    if kind == "clustered":
        centres = torch.randn((4096, d), generator=g, dtype=torch.float32, device=device)
        centres /= centres.norm(dim=1, keepdim=True)
        rows = gen_clustered_rows(torch, device, n, d, centres, 777)
        cq = centres[torch.randint(0, 4096, (pool,), generator=g, device=device)]
        nz = torch.randn((pool, d), generator=g, dtype=torch.float32, device=device)
        q = cq + 0.3 * nz / nz.norm(dim=1, keepdim=True)
        aimed = pool
    else:
        rows, starts = gen_near_duplicate_rows(torch, device, n, d, 4141)
        q = torch.randn((pool, d), generator=g, dtype=torch.float32, device=device)
        pick = starts[torch.randint(0, len(starts), (pool // 2,), generator=g, device=device)]
        nz = torch.randn((pool // 2, d), generator=g, dtype=torch.float32, device=device)
        q[0::2] = rows[pick + 5] + 0.02 * nz / nz.norm(dim=1, keepdim=True)  # every other query: next to a group's 6th member
        aimed = pool // 2
    q = (q / q.norm(dim=1, keepdim=True)).double().contiguous()
    torch.cuda.synchronize()
    index = DeviceIndex.from_device_ptr(rows.data_ptr(), n, d, local_rank, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    searcher = ShardedSearcher(local_index=index)
    steps = 12 if kind == "clustered" else 4
    flags = torch.zeros((16, B), dtype=torch.int32, device=device)
    index.profile(True)
    for i in range(PRECONDITION_STEPS if kind == "clustered" else 2):
        searcher.search(q[(i % 16) * B : (i % 16 + 1) * B], k, args.metric)
    torch.cuda.synchronize()
    index.profile_read(reset=True)
    t0 = time.perf_counter()
    for i in range(steps):
        out = searcher.search(q[i * B : (i + 1) * B], k, args.metric, out_flags=flags[i])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    index.profile(False)
    launches, scan_ms = index.profile_read(reset=True)
    exact = int((flags[:steps] // 2).sum().item())
    got_rows = out[1].cpu().numpy()
    got_cnt = out[2].cpu().numpy()
    stats = index.scan_stats() if hasattr(index, "scan_stats") else None
    index.close()
    host = rows.cpu().numpy()
    del rows
    torch.cuda.empty_cache()
    qh = q[(steps - 1) * B : steps * B].cpu().numpy()
    t0 = time.perf_counter()
    same = True
    for i in (0, 1):  # (near_duplicate: query 0 is aimed at a group, query 1 is not)
        want, _ = oracle_index.find_flat(qh[i], host, args.metric, k)
        same &= bool(np.array_equal(got_rows[i, : got_cnt[i]], want))
    t_oracle = time.perf_counter() - t0
    del host
    d_pad = (d + 127) // 128 * 128
    int8 = bool(stats and stats.get("int8_first_stage") and k <= 16)
    bytes_launch = (n * d_pad * sieve_image_bytes_per_element(int8) + (0 if args.metric == "inner_product" else 4 * n) + (n // 32 * 16 if int8 else 0)
                    + B * d * 4 + B * k * 12)  # what the sieve moves
    avg_ms = scan_ms / max(launches, 1)
    res = {"workload": f"{kind}: {n} x {d} float32 unit rows, {args.metric}, k={k}, {B} queries per step, {steps} steps of fresh queries",
           "ms_per_step": round(1e3 * el / steps, 4), "qps": round(B * steps / el, 1),
           "scan_bracket_ms": round(avg_ms, 4),
           "hbm_frac": round(bytes_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "mfma_frac": round(2.0 * n * d_pad * B / (avg_ms * 1e-3) / 1e12 / (MFMA_PEAK_TOPS_I8 if int8 else MFMA_PEAK_TFLOPS_BF16), 4),
           "first_stage": "int8" if int8 else "bf16",
           "exact_pass_queries": exact, "exact_pass_share": round(exact / (steps * B), 4),
           "queries_aimed_at_clusters_share": round(aimed / pool, 2),
           "ids_identical_to_cpu_oracle_on_2_queries": same, "oracle_s": round(t_oracle, 1)}
    if stats:
        res["scan_stats"] = stats
    return res


BM25_VOCAB = 50_000


def gen_bm25_corpus(np, torch, device, n_docs, seed):
    """SURVEY.md 8(d) BM25 corpus at token-id level: doc length clip(round(N(150, 40)), 1, 400), 0.1 % empty documents,
    term ids Zipf(s = 1.07) truncated to the 50k vocabulary (ids past it fold onto the last one, as
    `minimum(zipf - 1, vocab - 1)` does).  Sampled on the GPU by inverse CDF (numpy's rejection sampler needs ~10 s per
    150M tokens on the host); returned as host arrays, which is what mir_bm25_create takes."""
    from scipy.special import zeta

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.clamp(torch.round(torch.normal(150.0, 40.0, (n_docs,), generator=g, device=device)), 1, 400).to(torch.int64)
    lens[torch.rand(n_docs, generator=g, device=device) < 0.001] = 0
    indptr = torch.zeros(n_docs + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(lens, 0)
    total = int(indptr[-1].item())
    k = torch.arange(1, BM25_VOCAB, dtype=torch.float64, device=device)
    cdf = torch.cumsum(k.pow(-1.07), 0) / float(zeta(1.07, 1))  # P(X <= k), k = 1 .. vocab-1; the rest is the last id
    toks = torch.empty(total, dtype=torch.int32, device=device)
    for a in range(0, total, 1 << 26):
        u = torch.rand(min(1 << 26, total - a), generator=g, device=device, dtype=torch.float64)
        toks[a : a + len(u)] = torch.searchsorted(cdf, u).to(torch.int32)  # 0-based id = rank - 1; u > cdf[-1] -> vocab-1
    return indptr.cpu().numpy(), toks.cpu().numpy()


def bm25_queries(np, nq, seed):
    """SURVEY.md 8(d) query mix: 2-8 terms, half from the mid-frequency band, 5 % with an out-of-vocabulary id, 5 %
    with a repeated term."""
    qr = np.random.default_rng(seed)
    out = []
    for i in range(nq):
        L = int(qr.integers(2, 9))
        q = [int(t) for t in (qr.integers(50, 5000, L) if i % 2 else qr.integers(0, BM25_VOCAB, L))]
        if i % 20 == 3:
            q[0] = BM25_VOCAB + 7
        if i % 20 == 5:
            q.append(q[0])
        out.append(q)
    return out


def bm25_leg(np, torch, args, local_rank):
    """BASELINE config 3: BM25 scoring + top-k over 1M chunks / 50k-term vocabulary on one GPU, queries and results
    resident in HBM (mir_bm25_search_device on the current stream)."""
    from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25

    device = torch.device("cuda", local_rank)
    t0 = time.perf_counter()
    indptr, toks = gen_bm25_corpus(np, torch, device, args.bm25_docs, 777)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev = DeviceBM25.from_token_ids(indptr, toks, BM25_VOCAB, device=local_rank)
    t_build = time.perf_counter() - t0
    info = dev.info()
    df, _, _, _ = dev.corpus_stats()
    stream = torch.cuda.current_stream().cuda_stream
    k = args.k
    runs = []
    for B in (1, 64, 4096):
        # B = 1: 64 DIFFERENT queries of the mix, one call each (a single query re-issued 2048 times - round 2's figure - was
        # whatever that one query happened to be: a frequent-term query of 201k postings, 12 x the mix's mean)
        qs = bm25_queries(np, 64 if B == 1 else B, 778 + (64 if B == 1 else B))
        calls = []
        for grp in ([[q] for q in qs] if B == 1 else [qs]):
            flat = torch.tensor(np.concatenate([np.asarray(q, np.int32) for q in grp]), dtype=torch.int32, device=device)
            ptr = torch.tensor(np.concatenate(([0], np.cumsum([len(q) for q in grp]))), dtype=torch.int32, device=device)
            calls.append((flat, ptr))
        o_idx = torch.zeros((B, k), dtype=torch.int64, device=device)
        o_sc = torch.zeros((B, k), dtype=torch.float64, device=device)
        o_cnt = torch.zeros(B, dtype=torch.int32, device=device)
        ws = torch.zeros((dev.workspace_bytes(B, k) + 7) // 8, dtype=torch.int64, device=device)
        call = lambda i: dev.search_device(calls[i % len(calls)][0].data_ptr(), calls[i % len(calls)][1].data_ptr(), B, k,  # noqa: E731
                                           o_idx.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), ws.data_ptr(), stream)
        reps = max(3, 2048 // B)
        for i in range(3):
            call(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            call(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        touched = float(np.mean([sum(int(df[t]) for t in q if 0 <= t < BM25_VOCAB) for q in qs]))
        runs.append({"queries_per_step": B, "qps": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                     "postings_per_query": round(touched, 1), "algorithmic_bytes_per_query": round(12 * touched, 1),
                     "postings_GBps": round(12 * touched * B / dt / 1e9, 2),
                     "frac_of_hbm_peak": round(12 * touched * B / dt / 1e9 / HBM_PEAK_GBS, 5)})
        if B == 1:
            runs[-1]["note"] = "mean over 64 different single-query calls of the mix (back to back on one stream)"
    res = {
        "workload": f"BM25Okapi top-{k} over {args.bm25_docs} synthetic chunks, {BM25_VOCAB}-term vocabulary (Zipf 1.07), "
                    "SURVEY 8(d) query mix, float64 scores",
        "postings": info["n_postings"], "index_hbm_bytes": info["hbm_bytes"],
        "build_s": {"generate_on_gpu": round(t_gen, 2), "device_build_incl_upload": round(t_build, 2)},
        "runs": runs,
        # what bounds it: per (tile, query) a workgroup adds ~postings/ntiles float64 values into LDS one term after the
        # other (a barrier per term) and ranks the touched documents; HBM is far from busy (frac_of_hbm_peak above)
        "bound": "lds+latency (per-term LDS accumulation and block top-k, not HBM)",
        "bound_evidence": "profiles/r04_bm25_pmc.md",
    }
    return res, dev, (indptr, toks)


def bm25_cpu_baseline(np, corpus, dev_full, args, local_rank):
    """The reference's own BM25 path - rank-bm25's dict loops (restated in oracle/bm25.py: BM25Okapi.__init__ and
    get_scores, then the reversed stable argsort) - on ONE host core over the first documents of the same corpus,
    and the GPU's bit-identity on that sample."""
    from aidial_rag_amd.retrievers.bm25_retriever import DeviceBM25
    from oracle import bm25 as ob

    indptr, toks = corpus
    n = min(20_000, len(indptr) - 1)
    ip, tk = indptr[: n + 1], toks[: indptr[n]]
    docs = [tk[ip[i] : ip[i + 1]].tolist() for i in range(n)]
    t0 = time.perf_counter()
    o = ob.BM25Okapi(docs)
    t_build = time.perf_counter() - t0
    qs = bm25_queries(np, 48, 991)
    t0 = time.perf_counter()
    want = [o.get_scores(q) for q in qs]
    tops = [ob.top_n_indexes(w, args.k) for w in want]
    dt = time.perf_counter() - t0
    dev = DeviceBM25.from_token_ids(ip, tk, BM25_VOCAB, device=local_rank)
    idx, sc, cnt = dev.search(qs, args.k)
    same = all(np.array_equal(idx[i], tops[i]) and np.array_equal(sc[i], want[i][tops[i]]) for i in range(len(qs)))
    dev.close()
    qps = len(qs) / dt
    return {"value": round(qps, 3), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"{len(qs)} queries over the first {n} of {len(indptr) - 1} documents, rank-bm25 0.2.2 dict loops restated "
                      f"(model build {t_build:.2f} s per REQUEST in the reference); cost is linear in documents: "
                      f"~{qps * n / (len(indptr) - 1):.3f} queries/s at full size",
            "gpu_scores_and_topk_bit_identical_on_sample": bool(same)}


def encoder_cpu_baseline(np, torch):
    """The reference's CPU encoder path (sentence-transformers -> transformers BertModel, embeddings.py:52-66; its
    production backend is OpenVINO, absent here) as float32 torch eager on all host cores: one outer batch of 128
    chunks (embeddings.py:24-26), padded, CLS + L2 normalise."""
    from oracle import encoder as oe

    model = oe.make_model(layers=12, seed=0)
    rng = np.random.default_rng(99)
    lens = np.clip(np.round(rng.normal(220, 60, 128)), 8, 512).astype(np.int64)
    L = int(lens.max())
    ids = torch.zeros((128, L), dtype=torch.long)
    mask = torch.zeros((128, L), dtype=torch.long)
    for i, n in enumerate(lens):
        ids[i, :n] = torch.from_numpy(rng.integers(999, 30522, n))
        mask[i, :n] = 1
    with torch.no_grad():
        model(input_ids=ids[:8], attention_mask=mask[:8])  # warm-up
        t0 = time.perf_counter()
        out = model(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0]
        out = out / out.norm(dim=1, keepdim=True)
        dt = time.perf_counter() - t0
    return {"value": round(128 / dt, 2), "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"one outer batch of 128 synthetic chunks (mean {lens.mean():.0f} tokens, padded to {L}), transformers "
                      "BertModel float32, torch eager (the reference's OpenVINO backend is not installed)"}


def build_embeddings_leg(np, torch, args, local_rank, tmpdir):
    """The index-build rate THROUGH THE PRODUCT SURFACE: texts -> `build_embeddings` (embeddings.py:102-108: outer batches
    of 128, WordPiece tokenisation on the host, shared encoder passes) -> List[np.ndarray].  Synthetic WordPiece
    vocabulary (no real one exists offline), random-init weights."""
    import asyncio
    import json as js

    from aidial_rag_amd.embeddings import embeddings as emb
    from aidial_rag_amd.embeddings.wordpiece import WordPieceTokenizer

    os.makedirs(tmpdir, exist_ok=True)
    rng = np.random.default_rng(7)
    letters = "abcdefghijklmnopqrstuvwxyz"
    words = ["".join(rng.choice(list(letters), rng.integers(2, 9))) for _ in range(20000)]
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(letters) + \
            ["##" + c for c in letters] + sorted(set(words))
    open(os.path.join(tmpdir, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
    js.dump({"tokenizer_class": "BertTokenizer", "do_lower_case": True, "model_max_length": 512},
            open(os.path.join(tmpdir, "tokenizer_config.json"), "w"))
    tok = WordPieceTokenizer.from_vocab_file(os.path.join(tmpdir, "vocab.txt"))  # what BgeEncoder.from_pretrained_dir installs
    enc = emb.BgeEncoder.from_state_dict(random_bge_small_state_dict(np), tokenizer=tok, device=local_rank)
    emb.set_bge_embedding_impl(enc)
    try:
        n = args.encode_chunks
        lens = np.clip(np.round(rng.normal(218, 60, n)), 6, 510).astype(np.int64)  # words ~ tokens with this vocabulary
        texts = [" ".join(rng.choice(words, L)) for L in lens]
        asyncio.run(emb.build_embeddings(texts[:256]))  # warm-up
        rates = []
        for _ in range(3):  # the first full-size call also allocates the encoder's workspaces: the steady state is reported
            p0 = enc._doc_commit().passes
            t0 = time.perf_counter()
            out = asyncio.run(emb.build_embeddings(texts))
            dt = time.perf_counter() - t0
            rates.append(round(n / dt, 1))
        t1 = time.perf_counter()
        ids = tok(texts[:1024], add_special_tokens=True, truncation=True, max_length=512)["input_ids"]
        t_tok = (time.perf_counter() - t1) / 1024
        res = {"chunks_per_s": rates[-1], "chunks_per_s_each_run": rates, "chunks": n, "outer_batch": emb.EMBEDDINGS_BATCH_SIZE,
               "route": ("streamed: one tokeniser thread ahead of one encoder caller, slabs of %d texts" % emb.STREAM_SLAB) if n > emb.STREAM_MIN_TEXTS
                        else "group commit: %d shared encoder passes" % (enc._doc_commit().passes - p0),
               "mean_tokens_per_chunk": round(float(np.mean([len(x) for x in ids])), 1),
               "host_tokenise_us_per_chunk_1thread": round(t_tok * 1e6, 1),
               "note": "texts in, List[np.ndarray] out; includes WordPiece tokenisation (native, mir_wordpiece_encode) and result hand-over",
               "ok": bool(len(out) == n)}
        res["both_builders"] = both_builders_leg(np, asyncio, emb, words, rng, n)
        return res
    finally:
        emb.set_bge_embedding_impl(None)
        enc.close()


def prose_chunks(np, words, rng, n):
    """Synthetic chunk TEXTS for the keyword builder (and, in the combined leg, for the encoder too): sentences of 6-22 words of
    the synthetic vocabulary with capitals, commas, periods, a few contractions, quotes and stopwords, ~1000 characters per
    chunk (the product's chunk size) - prose-shaped input for word_tokenize; the words themselves mean nothing."""
    stop = ["the", "of", "and", "a", "to", "in", "is", "that", "it", "was", "for", "on", "are", "as", "with", "his", "they", "be", "at", "this"]
    extra = ["isn't", "don't", "it's", "we're", "cannot", "U.S.", "e.g.", "1,000", "3.5%", "(see", "above)", '"quoted"', "--"]
    out = []
    for _ in range(n):
        parts, length, target = [], 0, int(rng.integers(800, 1100))
        while length < target:
            k = int(rng.integers(6, 23))
            ws = [str(w) for w in rng.choice(words, k)]
            for i in range(k):
                r = rng.random()
                if r < 0.30:
                    ws[i] = stop[int(rng.integers(len(stop)))]
                elif r < 0.33:
                    ws[i] = extra[int(rng.integers(len(extra)))]
                if rng.random() < 0.08 and i + 1 < k:
                    ws[i] += ","
            ws[0] = ws[0][:1].upper() + ws[0][1:]
            sent = " ".join(ws) + (".", ".", ".", "?", "!")[int(rng.integers(5))]
            parts.append(sent)
            length += len(sent) + 1
        out.append(" ".join(parts))
    return out


def both_builders_leg(np, asyncio, emb, words, rng, n):
    """Index build through BOTH builders of the product surface, as load_document_impl runs them in one TaskGroup
    (documents.py:188-198): `BM25Retriever.build_index` (bm25_retriever.py:106-114: keywords_preprocess per chunk ->
    `tokenized_text`) and `build_embeddings` (embeddings.py:102-108) over the same chunk texts - each alone, then together."""
    from aidial_rag_amd import keywords_search as ks
    from aidial_rag_amd.retrievers.bm25_retriever import BM25Retriever

    class Chunk:  # what the builders read of a chunk (document_loaders' Chunk.text)
        __slots__ = ("text",)

        def __init__(self, t):
            self.text = t

    texts = prose_chunks(np, words, rng, n)
    chunks = [Chunk(t) for t in texts]
    asyncio.run(BM25Retriever.build_index(chunks[:512]))  # warm-up (front-end choice, thread start)
    rates = {"bm25": [], "embeddings": [], "together": []}
    items = None
    for _ in range(2):
        t0 = time.perf_counter()
        items = asyncio.run(BM25Retriever.build_index(chunks))
        rates["bm25"].append(round(n / (time.perf_counter() - t0), 1))
        t0 = time.perf_counter()
        embs = asyncio.run(emb.build_embeddings(texts))
        rates["embeddings"].append(round(n / (time.perf_counter() - t0), 1))

        async def both():
            return await asyncio.gather(BM25Retriever.build_index(chunks), emb.build_embeddings(texts))

        t0 = time.perf_counter()
        it2, em2 = asyncio.run(both())
        rates["together"].append(round(n / (time.perf_counter() - t0), 1))
    t0 = time.perf_counter()
    ref = [ks.keywords_preprocess(t) for t in texts[:64]]  # the per-chunk Python mirror (what round 3's build_index ran, one thread)
    t_py = (time.perf_counter() - t0) / 64
    return {"through_bm25_build_index_chunks_per_s": rates["bm25"][-1], "through_build_embeddings_chunks_per_s": rates["embeddings"][-1],
            "both_builders_together_chunks_per_s": rates["together"][-1], "each_run": rates, "chunks": n,
            "mean_characters_per_chunk": round(float(np.mean([len(t) for t in texts])), 1),
            "mean_keyword_tokens_per_chunk": round(float(np.mean([len(i.tokenized_text) for i in items])), 1),
            "host_threads": os.cpu_count(), "front_end": ks.front_end_info()["front_end"],
            "python_mirror_one_thread_chunks_per_s": round(1.0 / t_py, 1),
            "native_equals_python_mirror_on_64_chunks": bool(all(a == b.tokenized_text for a, b in zip(ref, items))),
            "note": "texts in -> TextIndexItem.tokenized_text out (TokenList views: list-like, a str object per DISTINCT token of a "
                    "4096-chunk batch, none per token); mir_keywords_preprocess + mir_kwp_result_dedupe on at most 16 host threads, "
                    "two batches in flight; `together` = asyncio.gather of both builders (documents.py:188-198), bounded by "
                    "build_embeddings (streamed slabs: a tokeniser thread and an encoder caller) sharing the interpreter with this one",
            "ok": bool(len(items) == n and len(embs) == n and len(it2) == n and len(em2) == n)}


def bm25_grouping_on_device(np, torch, device, indptr, toks, vocab):
    """`oracle.bm25.group_postings` - the INTEGER bookkeeping of the CSR restatement: distinct (term, document) pairs
    grouped by term, their term frequencies, document frequencies, first positions - with the sort done by torch on the
    GPU (the numpy sort needs minutes at 1e9 tokens).  Checker plumbing only: every float64 operation of the oracle
    stays in oracle/bm25.py.  Equality with the oracle's own numpy grouping: tests/test_gpu_sharded_c4.py."""
    n = len(indptr) - 1
    t = torch.from_numpy(np.ascontiguousarray(toks)).to(device)
    lens = torch.from_numpy(np.diff(indptr)).to(device)
    doc = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=device), lens)
    key = t.to(torch.int64) * n + doc
    del doc
    first = torch.full((vocab,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=device)
    first.scatter_reduce_(0, t.to(torch.int64), torch.arange(len(toks), dtype=torch.int64, device=device), reduce="amin")
    del t
    key, _ = torch.sort(key)
    ukey, tf = torch.unique_consecutive(key, return_counts=True)
    del key
    out = ((ukey % n).cpu().numpy(), tf.cpu().numpy().astype(np.int64),
           torch.bincount(ukey // n, minlength=vocab).cpu().numpy().astype(np.int64), first.cpu().numpy())
    del ukey, tf
    torch.cuda.empty_cache()
    return out


HYBRID_TERMS = 3   # keyword terms per hybrid query
HYBRID_BATCHES = 32  # distinct query batches: every preconditioning and timed step gets fresh queries


def hybrid_leg(np, torch, dist, args, world, rank, local_rank, barrier, DeviceIndex, ShardedSearcher, check):
    """BASELINE config 4: hybrid semantic + BM25 + fusion, chunks sharded by row / document across the GPUs (weak
    scaling: --hybrid-docs chunks per GPU; 8 GPUs x 1.25M = the 10M chunks of C4).  A step = B queries through the
    vector leg (scan + exact re-score + all-gather + merge), the BM25 leg (per-shard scoring + top-k, all-gather,
    merge with the reversed tie-break) and the host fusion of the two k = 7 lists (retrieval_chain.py:203-245).
    The two legs index the SAME chunks and a query asks both about the same thing: its vector lies near a target
    chunk's embedding and its keywords are that chunk's three rarest terms, so the legs' results overlap and the
    fusion's score-summing / de-duplication runs.  Every step takes a fresh batch of queries.  `check` (one GPU): the
    last batch's first queries against the unsharded oracle pipeline (oracle.find_flat + oracle.bm25 + oracle.fusion)."""
    from aidial_rag_amd.retrievers.sharded_bm25 import ShardedBM25, ShardedHybrid

    device = torch.device("cuda", local_rank)
    n_loc, d, B, k = args.hybrid_docs, args.dim, args.batch, 7
    lo = rank * n_loc
    g = torch.Generator(device=device)
    g.manual_seed(555 + rank)
    rows = torch.randn((n_loc, d), generator=g, dtype=torch.float32, device=device)
    rows /= rows.norm(dim=1, keepdim=True)
    index = DeviceIndex.from_device_ptr(rows.data_ptr(), n_loc, d, local_rank, row_offset=lo,
                                        stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    indptr, toks = gen_bm25_corpus(np, torch, device, n_loc, 9000 + rank)
    t0 = time.perf_counter()
    kw = ShardedBM25.build(indptr, toks, BM25_VOCAB, doc_offset=lo, device_index=local_rank)
    t_build = time.perf_counter() - t0
    # ---- the query pool: every rank fills in the queries whose target chunk it owns, one all-reduce shares them ----
    nq = HYBRID_BATCHES * B
    targets = np.random.default_rng(4321).integers(0, n_loc * world, nq)
    mine = np.flatnonzero((targets >= lo) & (targets < lo + n_loc))
    g.manual_seed(977)
    noise = torch.randn((nq, d), generator=g, dtype=torch.float32, device=device)
    qv = torch.zeros((nq, d), dtype=torch.float64, device=device)
    qt = torch.zeros((nq, HYBRID_TERMS), dtype=torch.int32, device=device)
    if len(mine):
        loc = torch.from_numpy(targets[mine] - lo).to(device)
        at = torch.from_numpy(mine).to(device)
        qv[at] = (rows[loc] + 0.05 * noise[at] / noise[at].norm(dim=1, keepdim=True)).double()
        freq = np.bincount(toks, minlength=BM25_VOCAB)
        terms = np.full((len(mine), HYBRID_TERMS), BM25_VOCAB + 7, np.int32)  # shorter chunks: padded with an unknown id (adds 0)
        for j, t in enumerate(targets[mine] - lo):
            u = np.unique(toks[indptr[t] : indptr[t + 1]])
            u = u[np.argsort(freq[u], kind="stable")][:HYBRID_TERMS]
            terms[j, : len(u)] = u
        qt[at] = torch.from_numpy(terms).to(device)
    if world > 1:
        dist.all_reduce(qv)
        dist.all_reduce(qt)
    sample = None
    if check:  # host copies for the oracle, before the device copies go
        sample = (rows.cpu().numpy(), indptr, toks)
    del rows, noise
    if not check:
        del indptr, toks
    hy = ShardedHybrid(ShardedSearcher(local_index=index), kw, k=k)
    ptr = torch.arange(0, HYBRID_TERMS * B + 1, HYBRID_TERMS, dtype=torch.int32, device=device)
    flats = [qt[i * B : (i + 1) * B].reshape(-1).contiguous() for i in range(HYBRID_BATCHES)]
    qvs = [qv[i * B : (i + 1) * B].contiguous() for i in range(HYBRID_BATCHES)]
    steps = HYBRID_BATCHES - PRECONDITION_STEPS

    def step(i):
        return hy.search(qvs[i], args.metric, flats[i], ptr)

    for i in range(PRECONDITION_STEPS):
        out = step(i)
    barrier()
    t0 = time.perf_counter()
    overlap = []
    for i in range(steps):
        out = step(PRECONDITION_STEPS + i)
        overlap.append(out)
    barrier()
    el = time.perf_counter() - t0
    tm = torch.tensor([el], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    el = float(tm.item())
    shared, fused_n = [], []
    for fused_ids, _, fused_cnt, v, t in overlap:
        shared += [len(set(v[0][i, : v[1][i]]) & set(t[0][i, : t[1][i]])) for i in range(B)]
        fused_n.append(float(fused_cnt.mean()))
    res = {"workload": f"hybrid: {n_loc} chunks per GPU x {world} GPUs = {n_loc * world} chunks; {d}-d float32 rows ({args.metric}) + "
                       f"BM25 ({BM25_VOCAB}-term vocabulary) over the same chunks; a query = a vector near a target chunk + that chunk's "
                       f"{HYBRID_TERMS} rarest terms; k = {k} per leg, reciprocal-rank fusion (weights 1, c = 60)",
           "scaling": "weak", "queries_per_step": B, "steps": steps, "fresh_queries_every_step": True,
           "ms_per_step": round(1e3 * el / steps, 4), "qps": round(B * steps / el, 1),
           "bm25_sharded_build_s": round(t_build, 2),
           "fused_results_per_query_mean": round(float(np.mean(fused_n)), 2),
           "legs_overlap_in_results_mean": round(float(np.mean(shared)), 3)}
    if check:
        res["oracle_check"] = hybrid_oracle_check(np, torch, device, sample, qv[-B:].cpu().numpy(), qt[-B:].cpu().numpy(), overlap[-1], args.metric, k)
    index.close()
    kw.model.close()
    return res


def hybrid_oracle_check(np, torch, device, sample, qvecs, qterms, got, metric, k, n_check=6):
    """The last timed hybrid step's first queries against the unsharded oracle pipeline on the same chunks."""
    from oracle import bm25 as ob
    from oracle import embeddings_index as oi
    from oracle import fusion as of

    rows, indptr, toks = sample
    t0 = time.perf_counter()
    model = ob.BM25OkapiCSR(indptr, toks, BM25_VOCAB, grouped=bm25_grouping_on_device(np, torch, device, indptr, toks, BM25_VOCAB))
    ids, scores, cnt, v, t = got
    ok_v = ok_t = ok_f = True
    for i in range(n_check):
        sem, _ = oi.find_flat(qvecs[i], rows, metric, k)
        top = ob.top_n_indexes(model.get_scores([int(x) for x in qterms[i]]), k)
        lists = [[int(x) for x in sem], [int(x) for x in top]]
        fused = of.weighted_reciprocal_rank(lists, [1.0, 1.0])
        fs = of.rrf_scores(lists, [1.0, 1.0])
        ok_v &= list(v[0][i, : v[1][i]]) == lists[0]
        ok_t &= list(t[0][i, : t[1][i]]) == lists[1]
        ok_f &= list(ids[i, : cnt[i]]) == fused and bool(np.array_equal(scores[i, : cnt[i]], np.asarray([fs[x] for x in fused])))
    return {"queries_checked": n_check, "vector_leg_ids_identical": bool(ok_v), "bm25_leg_ids_identical": bool(ok_t),
            "fused_ids_and_scores_identical": bool(ok_f), "oracle_s": round(time.perf_counter() - t0, 1)}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process touches no GPU, starts N ranks under
    `torch.distributed.run` (one per GPU, RCCL over 127.0.0.1), relays their output and exits with their code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    raise SystemExit(proc.wait())


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    from aidial_rag_amd import _native as nat
    from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
    from aidial_rag_amd.retrievers.sharded_index import ShardedSearcher, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 code path on a box with ONE GPU (every rank on device 0, collectives over gloo): exercises the
    # sharding, the gathered blobs and the merge kernels end to end; its numbers mean nothing.
    rehearsal = os.environ.get("MIR_BENCH_ONE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if nat.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    n, d, B, k = args.rows, args.dim, args.batch, args.k
    lo, hi = shard_bounds(n, world, rank)

    # ---- index build (untimed here; reported as a side number) ----
    t0 = time.time()
    shard = gen_rows(torch, device, lo, hi, d)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    index = DeviceIndex.from_device_ptr(shard.data_ptr(), hi - lo, d, local_rank, row_offset=lo,
                                        stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    sample_rows = min(args.cpu_rows, hi - lo) if rank == 0 and world == 1 else 0
    sample = shard[:sample_rows].cpu().numpy() if sample_rows else None
    del shard
    torch.cuda.empty_cache()

    g = torch.Generator(device=device)
    g.manual_seed(4321)
    nq_pool = max(4096, B)
    q32 = torch.randn((nq_pool, d), generator=g, dtype=torch.float32, device=device)
    q32 /= q32.norm(dim=1, keepdim=True)
    queries = q32.double().contiguous()  # the live path hands float64 queries (semantic_retriever.py:49,53)

    searcher = ShardedSearcher(local_index=index)
    # Steps are independent batches; a server keeps several in flight.  The timed loop issues them round-robin on
    # --streams HIP streams (each with its own searcher, i.e. its own result buffers and library workspace), so one step's
    # short dependent kernels (verify / select / gate, the all-gather and merge at N > 1) run beside the next step's
    # streaming launches: 1.41 -> 1.39 ms per step at 10M rows, 0.257 -> 0.216 ms on a 1.25M-row shard (tools/two_stream_steps.py).
    n_streams = args.streams if args.streams > 0 else (1 if hi - lo >= 4_000_000 else 3)
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream(device)]
    searchers = [searcher] + [ShardedSearcher(local_index=index) for _ in range(n_streams - 1)]

    # every step's completeness flags land in their own row: no per-step reduction kernels inside the timed loop
    flags_all = torch.zeros((max(args.warmup, 1) + args.steps, B), dtype=torch.int32, device=device)

    def step(i):
        s = (i * B) % (nq_pool - B + 1)
        with torch.cuda.stream(streams[i % n_streams]):
            return searchers[i % n_streams].search(queries[s : s + B], k, args.metric, out_flags=flags_all[i])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    index.profile(True)  # before the warm-up: event creation is slow and must not be timed
    torch.cuda.synchronize()  # (the side streams start after everything the default stream has prepared)
    n_warm = max(args.warmup, 1)
    # Preconditioning, before the W warm-up steps and untimed like them: from idle the GPU needs ~10 launches
    # (~30 ms) to reach its steady clock / power state - measured with --steps 5: scan 3.48 ms per launch after 2
    # warm-up steps, 3.17 after 5, 3.02 after 8, 2.90 after 12, 2.89 after 20 - so a short --warmup would time the ramp.
    precondition = max(0, PRECONDITION_STEPS - n_warm)
    for i in range(precondition):
        step(i % n_warm)
    for i in range(n_warm):
        out = step(i)  # exactly the timed loop's body
    barrier()
    index.profile_read(reset=True)
    index.scan_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(n_warm + i)
    barrier()
    elapsed = time.perf_counter() - t0
    flags_total = flags_all[n_warm:].sum(dtype=torch.int64).reshape(())
    # the LAST timed step's own answers (not a separate index): compared with the CPU oracle over ALL rows further down
    last_s = ((n_warm + args.steps - 1) * B) % (nq_pool - B + 1)
    last_rows, last_cnt = out[1][:MAIN_LEG_CHECK_QUERIES].cpu().numpy(), out[2][:MAIN_LEG_CHECK_QUERIES].cpu().numpy()
    last_q = queries[last_s : last_s + MAIN_LEG_CHECK_QUERIES].cpu().numpy()
    launches, scan_ms = index.profile_read(reset=True)
    sieve_stats = index.scan_stats(reset=True)
    bracket_over = "the timed region"
    if n_streams > 1:
        # with several steps in flight a launch's HIP-event bracket also holds the kernels other streams ran beside it: the
        # dominant kernel's duration is taken from min(K, 10) more steps of the same loop on ONE stream, right after the timed region
        extra = min(args.steps, 10)
        with torch.cuda.stream(streams[0]):
            for i in range(extra):
                searchers[0].search(queries[i * B % (nq_pool - B + 1) :][:B], k, args.metric, out_flags=flags_all[0])
        torch.cuda.synchronize()
        launches, scan_ms = index.profile_read(reset=True)
        index.scan_stats(reset=True)
        bracket_over = f"{extra} more steps on one stream right after the timed region (its {n_streams} streams overlap each other's brackets)"
    index.profile(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(flags_total, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    qps = B * args.steps / elapsed

    # ---- roofline of the dominant kernel, per shard pass ----
    # `bytes_per_launch` = what the dominant kernel's ALGORITHM moves per shard pass.  Since round 3 a float32 shard of >= 32K rows at
    # d <= 384 is searched by the sieve (csrc/vec_kernels_sieve.h), whose filter streams only the bf16 hi blocks of the index image:
    # rows * d_pad * 2 + the norm column + the query tile + its results (`traffic`, from the PMC counters, confirms it: 1.002 x).
    # SURVEY.md 8(d)'s figure for a float32 pass (rows * d * 4 + ...) is kept beside it as `survey_bytes`; the ratio of the two rates
    # is `speedup_vs_f32_stream_at_peak` (how many times faster the pass is than a float32 stream at 8 TB/s could be - NOT a fraction
    # of any peak).  At 256 queries per pass the filter is as much matrix-bound as stream-bound, so both fractions are computed and
    # `bound` / `achieved` / `peak` / `unit` / `frac` are those of whichever roof the launch is closer to; neither can exceed 1.
    n_loc = hi - lo
    aux = 0 if args.metric == "inner_product" else 4 * n_loc
    wide = d <= 384 and d > 64 and k <= 52
    sieve = d <= 384 and d > 64 and k <= 64 and n_loc >= 32768 and os.environ.get("MIR_NO_SIEVE") is None
    qpl = (256 if B > 128 else 128) if sieve else 128 if wide else 32
    q_launch = min(B, qpl)  # queries actually riding one launch
    passes = -(-B // qpl)   # launches groups per step (1 at the default batch)
    survey_bytes = n_loc * d * 4 + aux + q_launch * d * 4 + q_launch * k * 12
    d_pad = (d + 127) // 128 * 128
    # which first stage this index's searches run: the int8 image (finite rows of one norm, k <= 16) or the bf16 hi blocks
    int8 = bool(sieve and sieve_stats.get("int8_first_stage") and k <= 16)
    tile_params = (n_loc // 32) * 16 if int8 else 0  # a float4 per 32-row tile: scale, residual bound
    bytes_launch = (n_loc * d_pad * sieve_image_bytes_per_element(int8) + aux + tile_params + q_launch * d * 4 + q_launch * k * 12) if sieve else survey_bytes
    avg_ms = scan_ms / max(launches, 1)
    hbm_gbs = bytes_launch / (avg_ms * 1e-3) / 1e9
    # HBM traffic per launch comes from PMC counters, which cannot be collected inside this run (rocprofv3 --pmc
    # serialises kernels and needs its own passes): it is READ from the tracked summary of the last counter run on
    # the same shape and labelled as such (`traffic_source`); null when no summary matches.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if (tj.get("rows_per_launch") == n_loc and tj.get("dim") == d and bool(tj.get("sieve")) == sieve
                and tj.get("first_stage", "bf16") == ("int8" if int8 else "bf16")):
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run): " + str(tj.get("command", ""))
    if sieve and int8:
        kernel_name = ("sieve_i8_kernel: two filter launches over the int8 image (first 1/16 of the rows, then the rest; v_mfma_i32_16x16x64_i8, "
                       "exact integer dot products, a rigorous per-tile margin) with sieve_scatter_i8_kernel + sieve_select_kernel between "
                       "them (the k-th largest lower bound so far = the second launch's threshold)")
    elif sieve:
        kernel_name = ("sieve_q16_kernel: two filter launches over the bf16 hi blocks (first 1/16 of the tiles, then the rest) with "
                       "sieve_scatter_kernel + sieve_select_kernel between them (the k-th largest filter value so far, less the margin = "
                       "the second launch's threshold)")
    elif wide:
        kernel_name = "scan_topk_q16_kernel (2 launches per shard + list_threshold_kernel)"
    else:
        kernel_name = "scan_topk_kernel"
    mfma_tflops = 2.0 * n_loc * d_pad * q_launch / (avg_ms * 1e-3) / 1e12 if sieve else None
    hbm_frac = hbm_gbs / HBM_PEAK_GBS
    mfma_peak = MFMA_PEAK_TOPS_I8 if int8 else MFMA_PEAK_TFLOPS_BF16
    mfma_frac = None if mfma_tflops is None else mfma_tflops / mfma_peak
    by_mfma = mfma_frac is not None and mfma_frac > hbm_frac

    result = {
        "metric": "retrieval_qps_10Mx384",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": ("f32 (int8 MFMA filter with exact integer dot products and a rigorous error margin" if int8 else "f32 (bf16 MFMA filter with a rigorous error margin")
                 + ", every candidate the margin cannot exclude re-scored in f64 with the reference formula)",
        "data": "synthetic",
        "config": {
            "workload": f"brute-force top-k over {n}x{d} float32 unit-norm rows, {args.metric}, k={k}",
            "queries_per_step": B,
            "steps_in_flight": n_streams,
            "rows_per_gpu": n_loc,
            "parallelism": f"row-shard x{world}, all-gather of partial top-k",
        },
        "roofline": {
            "bound": "mfma" if by_mfma else "hbm",
            # HIP events on the launch stream bracket the shard pass: both filter launches and what runs between them.  The
            # 32K-row sample launch before them and the scatter / select kernels after the second launch are outside the
            # bracket and inside `step_hbm_frac`.
            "kernel": kernel_name,
            "queries_per_launch": qpl,
            "achieved": round(mfma_tflops if by_mfma else hbm_gbs, 1),
            "peak": mfma_peak if by_mfma else HBM_PEAK_GBS,
            "unit": ("TOP/s" if int8 else "TFLOP/s") if by_mfma else "GB/s",
            "frac": round(mfma_frac if by_mfma else hbm_frac, 4),
            "hbm_gbs": round(hbm_gbs, 1),
            "hbm_frac": round(hbm_frac, 4),
            "mfma_tflops_hi_hi": None if mfma_tflops is None else round(mfma_tflops, 1),  # (int8 first stage: TOP/s of the int8 products)
            "mfma_peak": mfma_peak if sieve else None,
            "mfma_frac": None if mfma_frac is None else round(mfma_frac, 4),
            "first_stage": ("int8" if int8 else "bf16") if sieve else None,
            "bytes_per_launch": bytes_launch,
            "bytes_per_launch_is": (("what the sieve's algorithm moves per shard pass: rows * d_pad * 1 (int8 image) + norm column + tile "
                                     "parameters + queries + results" if int8 else
                                     "what the sieve's algorithm moves per shard pass: rows * d_pad * 2 (bf16 hi blocks) + norm column + "
                                     "queries + results") if sieve else "SURVEY.md 8(d) algorithmic bytes of one shard pass"),
            "flops_per_launch": None if mfma_tflops is None else 2.0 * n_loc * d_pad * q_launch,
            "survey_bytes": survey_bytes,
            "survey_bytes_is": "SURVEY.md 8(d): rows * d * 4 + norm column + queries + results (a float32 stream of the shard)",
            "speedup_vs_f32_stream_at_peak": round(survey_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "avg_launch_ms": round(avg_ms, 4),
            "launches": launches,
            "measured_over": bracket_over,
            # the same bytes over the whole step (prep, threshold pre-pass, filter, scatter, select, exact-pass gate, merge)
            "step_hbm_frac": round(passes * bytes_launch / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
        },
        "sieve": sieve_stats if sieve else None,  # candidates per query and filter launch, queries handed to the exact pass (rank 0's shard)
        "exact_pass_queries": int(flags_total.item()) // 2,  # queries the filter could not prove (flag bit 2), recomputed exactly
        "preconditioning_steps": precondition,
        **({"rehearsal_one_gpu": True} if rehearsal else {}),
        "index_build_s": {"generate": round(t_gen, 2), "upload_pack_norms": round(t_build, 2)},
    }

    # ---- the same index at other batch sizes: queries per pass trade QPS against roofline fraction ----
    if args.sweep:
        sweep = []
        for Bs in (1, 32, 64, 128, 192, 256):
            if Bs == B:
                continue
            index.profile(True)
            for i in range(5):
                searcher.search(queries[i * Bs : (i + 1) * Bs], k, args.metric)
            barrier()
            index.profile_read(reset=True)
            t0 = time.perf_counter()
            for i in range(10):
                searcher.search(queries[i * Bs : (i + 1) * Bs], k, args.metric)
            barrier()
            el = time.perf_counter() - t0
            index.profile(False)
            ln, ms = index.profile_read(reset=True)
            tm = torch.tensor([el], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
            a_ms = ms / max(ln, 1)
            sweep.append({"queries_per_step": Bs, "qps": round(Bs * 10 / el, 1), "ms_per_step": round(1e3 * el / 10, 4),
                          "scan_launch_ms": round(a_ms, 4), "hbm_frac": round(bytes_launch / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "mfma_frac": (None if not sieve else
                                        round(2.0 * n_loc * d_pad * min(Bs, qpl) / (a_ms * 1e-3) / 1e12 / mfma_peak, 4))})
        result["batch_sweep"] = sweep
        # the other metrics of embeddings_metrics.py at the headline batch (north_star names cosine; the index is the same image)
        others = []
        for mt in ("cosine_sim", "inner_product", "euclidean_dist", "sqeuclidean_dist"):
            if mt == args.metric:
                continue
            for i in range(5):
                searcher.search(queries[i * B : (i + 1) * B], k, mt)
            barrier()
            t0 = time.perf_counter()
            for i in range(10):
                searcher.search(queries[i * B : (i + 1) * B], k, mt)
            barrier()
            el = time.perf_counter() - t0
            tm = torch.tensor([el], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
            others.append({"metric": mt, "queries_per_step": B, "qps": round(B * 10 / el, 1), "ms_per_step": round(1e3 * el / 10, 4)})
        result["metric_sweep"] = others
    index.close()  # release the shard before the other legs allocate theirs
    if world == 1 and rank == 0 and args.cpu_legs:
        # (after the timed region and the sweeps: the oracle is the checker, never the thing measured)
        result["timed_leg_check"] = timed_leg_check(np, torch, device, n, d, args, last_q, last_rows, last_cnt)
    if args.variants and world == 1:
        for kind in ("clustered", "near_duplicate"):
            result[kind] = variant_leg(np, torch, args, local_rank, DeviceIndex, ShardedSearcher, kind)
    if args.c5_rows > 0:
        result["c5_float16_d1024"] = c5_leg(np, torch, dist, args, world, rank, local_rank, barrier, DeviceIndex, ShardedSearcher)
    if args.encode_chunks > 0:
        result["index_build"] = encoder_leg(np, torch, dist, args, world, rank, local_rank, barrier)
        if rank == 0:
            import tempfile

            with tempfile.TemporaryDirectory() as td:
                result["index_build"]["through_build_embeddings"] = build_embeddings_leg(np, torch, args, local_rank, td)
    if args.c2_chunks > 0 and world == 1:
        result["c2_encode_index_search"] = c2_leg(np, torch, args, local_rank, DeviceIndex, ShardedSearcher)
    if args.hybrid_docs > 0:
        result["hybrid_c4"] = hybrid_leg(np, torch, dist, args, world, rank, local_rank, barrier, DeviceIndex, ShardedSearcher,
                                         check=(world == 1 and args.cpu_legs))
    bm25_state = None
    if args.bm25_docs > 0 and world == 1:
        result["bm25_c3"], bm25_dev, bm25_corpus = bm25_leg(np, torch, args, local_rank)
        bm25_state = (bm25_dev, bm25_corpus)
    if rank == 0 and world == 1 and sample is not None and len(sample):
        result["cpu_baseline"] = cpu_baseline(np, sample, queries[: args.cpu_queries].cpu().numpy(), args, DeviceIndex)
        if args.cpu_legs:
            # the other two legs of the reference's CPU path (north_star: "CPU embeddings+BM25 path timed in the same run")
            if bm25_state is not None:
                result["cpu_baseline"]["bm25"] = bm25_cpu_baseline(np, bm25_state[1], bm25_state[0], args, local_rank)
            result["cpu_baseline"]["encoder"] = encoder_cpu_baseline(np, torch)
    if bm25_state is not None:
        bm25_state[0].close()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def timed_leg_check(np, torch, device, n, d, args, qs, got_rows, got_cnt):
    """The first MAIN_LEG_CHECK_QUERIES queries of the LAST TIMED STEP (the headline shape: one 256-query launch over all
    --rows rows) against the CPU oracle over all rows, regenerated from the same seed."""
    from oracle import embeddings_index as oracle_index

    host = gen_rows(torch, device, 0, n, d).cpu().numpy()
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    same = True
    for i in range(len(qs)):
        want, _ = oracle_index.find_flat(qs[i], host, args.metric, args.k)
        same &= bool(np.array_equal(got_rows[i, : got_cnt[i]], want))
    return {"queries_of_the_last_timed_step_checked": len(qs), "rows": n, "ids_identical_to_cpu_oracle": same,
            "oracle_s": round(time.perf_counter() - t0, 1)}


def cpu_baseline(np, sample, qs, args, DeviceIndex):
    """The oracle (numpy restatement of embeddings_metrics.py + embeddings_index.py) on the host cores,
    float64 query as on the live path, full stable argsort - and a parity check of the GPU on the same rows."""
    from oracle import embeddings_index as oracle_index

    t0 = time.perf_counter()
    want = [oracle_index.find_flat(q, sample, args.metric, args.k) for q in qs]
    dt = time.perf_counter() - t0
    dev = DeviceIndex.from_host(sample)
    _, _, rows, dist_, cnt, flags = dev.search(qs, args.k, args.metric)
    ids_equal = all(np.array_equal(rows[i, : cnt[i]], want[i][0]) for i in range(len(qs)))
    max_err = max(float(np.max(np.abs(dist_[i, : cnt[i]] - want[i][1]))) for i in range(len(qs)))
    dev.close()
    qps = len(qs) / dt
    return {
        "value": round(qps, 4),
        "unit": "queries/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": f"{len(qs)} float64 queries over the first {len(sample)} of {args.rows} rows "
                  f"(cost is linear in rows: ~{qps * len(sample) / args.rows:.4f} queries/s at full size)",
        "gpu_ids_identical_on_sample": bool(ids_equal),
        "gpu_max_abs_dist_err_on_sample": max_err,
    }


if __name__ == "__main__":
    main()
