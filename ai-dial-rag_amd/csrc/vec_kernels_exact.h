// The exact pass, batched (round 3): the reference's own computation (embeddings_index.py:51-60: the metric in float64
// for EVERY row, then the stable order on (distance, row)) for the queries the filters hand over - a full candidate
// buffer of the sieve, an unproven query of the list scans, any k beyond their lists.
//
// Round 2's exact_topk_kernel walked the flagged queries one after the other, each a private pass over all rows with a
// wave per row (lane-strided partial sums, a 6-step wave reduction per row and query): 8.4 ms per query at 10M x 384,
// 128 flagged queries = 1.1 s.  Here every row is read ONCE for up to 128 flagged queries:
//   * ROWS ON LANES.  One workgroup of 16 waves per CU stages 64 rows x 192 columns in LDS (row stride 193 floats:
//     conflict-free column walks); lane l then owns row l and accumulates its dot products sequentially over the
//     columns - no cross-lane reduction at all.  The query values are the same for every lane: they come through the
//     scalar cache (s_load) from a transposed copy Qt[pass][column][128 queries] that the flagging kernels write, and
//     enter v_fma_f64 as the scalar operand.  Wave w takes queries 8w .. 8w + 7 of the pass: per staged element one LDS
//     read, one conversion and eight float64 FMAs, four columns (4 x 8 query values = 64 SGPRs) per scalar-memory
//     wait; the other three waves of its SIMD compute meanwhile.  The next slice is fetched into registers (48 bytes
//     per thread) before the current one is computed on.
//   * a wave keeps its 8 queries' best `kk` <= 64 entries in registers, one entry per lane, sorted (exact_wave_insert);
//     a block of 64 rows costs one vote per query unless a row actually enters a list.
//   * per pass the workgroups' lists meet in HBM and the last workgroup to arrive merges them (as round 2's kernel).
// float64 arithmetic: the reference's formulas (embeddings_metrics.py:14-50) with one fused multiply-add per element,
// summed in ascending column order.  (exact_metric_wave sums lane-strided partials and a tree; both are the
// reference's float64 formula to within its own rounding - numpy's BLAS order is not defined either - and every row
// of one query's answer is computed by the same routine.)  cosine_sim divides by the row norm that
// row_dnorm_kernel stores at build time with exact_metric_wave's own summation order.
// Device-gated: reads the number of flagged queries and exits at once when it is 0.
#pragma once
#include "vec_kernels.h"

namespace mir {

constexpr int kXbThreads = 1024;
constexpr int kXbWaves = 16;
constexpr int kXbRows = 64;      // rows per block = lanes
constexpr int kXbStride = kXbCols + 1;
constexpr int kXbQW = 8;         // queries per wave and pass
static_assert(kXbQ == kXbWaves * kXbQW, "queries per pass");

// max((float)sqrt(sum x^2), 1e-8) per row, with exact_metric_wave's summation order (lane-strided, then the xor tree)
template <typename T>
__global__ __launch_bounds__(256) void row_dnorm_kernel(const T *__restrict__ docs, int64_t n, int d, float *__restrict__ dnorm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const T *a = docs + row * (int64_t)d;
    constexpr int U = 8;
    double s = 0.0;
    for (int j0 = lane; j0 < d; j0 += 64 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = j0 + 64 * u < d ? (float)a[j0 + 64 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j0 + 64 * u < d) {
                const double x = (double)v[u];
                s += x * x;
            }
    }
    s = wave_sum(s);
    if (lane == 0) dnorm[row] = fmaxf((float)sqrt(s), 1e-8f);
}

struct ExactBatchArgs {
    const float *docs;         // f32 [n][d], or null with
    const _Float16 *docs16;    // f16 [n][d]
    const float *doc_sq;       // [n] (numpy float32 pairwise sum)
    const float *dnorm;        // [n] row_dnorm_kernel
    uint32_t n_rows;
    int d;
    int metric;
    const double *qt;          // [passes][dpad][32]
    const double *q_sq;
    const int32_t *nflag;
    const int32_t *flagged;
    int k;
    int round;
    int list_stride;           // min(k, kExactRound)
    uint64_t *part;            // [b][grid][list_stride][2]: {dist bits, valid << 32 | row}
    uint32_t *arrive;          // [b], zero between launches
    double *bound_dist;        // [b] last result of the previous round
    uint32_t *bound_row;       // [b]
    const int64_t *chunk_ids;
    const int32_t *doc_ids;
    int64_t row_offset;
    int32_t *out_doc;
    int64_t *out_chunk;
    int64_t *out_row;
    double *out_dist;
    int32_t *out_count;
    int32_t *out_flags;
};

// rank of this thread's entry among the workgroup's wave lists (WAVES lists of <= 64, sorted); -1: no entry
template <int WAVES>
__device__ __forceinline__ int xb_block_rank(double my_d, uint32_t my_r, int cnt, double *s_d, uint32_t *s_r, int *s_cnt, int tid,
                                             int *total_out) {
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    s_d[tid] = my_d;
    s_r[tid] = my_r;
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    int total = 0, rank = 0;
    const bool valid = lane < cnt;
    for (int w = 0; w < WAVES; ++w) {
        const int c = s_cnt[w];
        total += c;
        if (valid)
            for (int l = 0; l < c; ++l) rank += dist_before(s_d[w * 64 + l], s_r[w * 64 + l], my_d, my_r) ? 1 : 0;
    }
    *total_out = total;
    return valid ? rank : -1;
}

template <typename T, bool COS>
__device__ __forceinline__ void exact_topk_batch(const ExactBatchArgs &a, const int nf) {
    __shared__ float tile[kXbRows * kXbStride];  // 49 408 B
    __shared__ double s_d[kXbThreads];
    __shared__ uint32_t s_r[kXbThreads];
    __shared__ int s_cnt[kXbWaves];
    __shared__ int s_last[kXbQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = min(kExactRound, a.k - kExactRound * a.round);
    const uint32_t G = gridDim.x;
    const int d = a.d, dpad = xb_dpad(d), nslices = dpad / kXbCols;
    const T *docs = reinterpret_cast<const T *>(sizeof(T) == 2 ? (const void *)a.docs16 : (const void *)a.docs);
    const uint32_t nblocks = (a.n_rows + kXbRows - 1) / kXbRows;
    const bool bounded = a.round > 0;
    const bool vec_ok = (d & 3) == 0;

    for (int p0 = 0; p0 < nf; p0 += kXbQ) {
        const int np = min(kXbQ, nf - p0);          // queries of this pass
        // (constant address space: the copy is read-only for this kernel, which is what lets the loads be scalar)
        typedef const __attribute__((address_space(4))) double cdouble;
        cdouble *qt = (cdouble *)(a.qt + (size_t)(p0 / kXbQ) * dpad * kXbQ + wave * kXbQW);
        // this wave's queries: slots wave * 8 + i
        double my_d[kXbQW], qsq[kXbQW];
        uint32_t my_r[kXbQW];
        int cnt[kXbQW];
        bool live[kXbQW];
#pragma unroll
        for (int i = 0; i < kXbQW; ++i) {
            const int c = wave * kXbQW + i;
            live[i] = c < np;
            const int qi = live[i] ? a.flagged[p0 + c] : 0;
            my_d[i] = 0.0; my_r[i] = 0; cnt[i] = 0;
            qsq[i] = live[i] ? a.q_sq[qi] : 0.0;
        }
        const bool wave_live = wave * kXbQW < np;

        // A slice = 64 rows x 192 columns = 3072 pieces of 16 bytes, three per thread.  The NEXT slice is fetched into
        // registers before this one is computed on, and stored to LDS after it.
        constexpr int kIts = kXbRows * (kXbCols / 4) / kXbThreads;
        static_assert(kIts * kXbThreads == kXbRows * (kXbCols / 4), "slice pieces per thread");
        float nx[kIts][4];
        auto fetch = [&](uint32_t blk, int s) {
#pragma unroll
            for (int it = 0; it < kIts; ++it) {
                const int idx = it * kXbThreads + tid;
                const int r = idx / (kXbCols / 4), c4 = idx % (kXbCols / 4);
                const int col = s * kXbCols + c4 * 4;
                const uint32_t gr = blk * kXbRows + r;
                nx[it][0] = 0.f; nx[it][1] = 0.f; nx[it][2] = 0.f; nx[it][3] = 0.f;
                if (blk < nblocks && gr < a.n_rows) {
                    const T *src = docs + (size_t)gr * d + col;
                    if (vec_ok && col + 3 < d) {
                        if (sizeof(T) == 4) {
                            const float4 v = *reinterpret_cast<const float4 *>(src);
                            nx[it][0] = v.x; nx[it][1] = v.y; nx[it][2] = v.z; nx[it][3] = v.w;
                        } else {
                            const uint2 v = *reinterpret_cast<const uint2 *>(src);
                            const _Float16 *h = reinterpret_cast<const _Float16 *>(&v);
                            nx[it][0] = (float)h[0]; nx[it][1] = (float)h[1]; nx[it][2] = (float)h[2]; nx[it][3] = (float)h[3];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (col + u < d) nx[it][u] = (float)src[u];
                    }
                }
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int it = 0; it < kIts; ++it) {
                const int idx = it * kXbThreads + tid;
                const int r = idx / (kXbCols / 4), c4 = idx % (kXbCols / 4);
                float *dst = tile + r * kXbStride + c4 * 4;
                dst[0] = nx[it][0]; dst[1] = nx[it][1]; dst[2] = nx[it][2]; dst[3] = nx[it][3];
            }
        };
        fetch(blockIdx.x, 0);
        for (uint32_t blk = blockIdx.x; blk < nblocks; blk += G) {
            const uint32_t row0 = blk * kXbRows;
            const uint32_t row = row0 + lane;
            const bool row_ok = row < a.n_rows;
            double acc[kXbQW];
#pragma unroll
            for (int i = 0; i < kXbQW; ++i) acc[i] = 0.0;
            const float dn = (COS && row_ok) ? a.dnorm[row] : 1.0f;
            for (int s = 0; s < nslices; ++s) {
                __syncthreads();  // the previous slice has been consumed
                stash();
                __syncthreads();
                if (s + 1 < nslices) fetch(blk, s + 1);
                else fetch(blk + G, 0);
                if (!wave_live) continue;
                // ---- lane = row: sequential float64 dot products with this wave's 8 queries
                const float *mine = tile + lane * kXbStride;
                cdouble *qs = qt + (size_t)s * kXbCols * kXbQ;
                const int jn = (min(kXbCols, d - s * kXbCols) + 3) & ~3;  // (columns past d are staged as zeros, Qt is 0 there)
                for (int j = 0; j < jn; j += 4) {
                    // four columns per scalar-memory wait: 4 LDS reads + 4 x 8 query values (64 SGPRs), then 32 FMAs
                    float x[4];
                    double qv[4][kXbQW];
                    cdouble *qj = qs + (size_t)j * kXbQ;  // wave-uniform: scalar loads
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x[u] = mine[j + u];
#pragma unroll
                        for (int i = 0; i < kXbQW; ++i) qv[u][i] = qj[u * kXbQ + i];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float xx = x[u];
                        if (COS) xx = __fdiv_rn(xx, dn);
                        const double xd = (double)xx;
#pragma unroll
                        for (int i = 0; i < kXbQW; ++i) acc[i] = fma(xd, qv[u][i], acc[i]);
                    }
                }
            }
            if (!wave_live) continue;
            // ---- the metric, and the rows that enter a list
            const float dsq = (row_ok && (a.metric == MIR_METRIC_SQEUCLIDEAN_DIST || a.metric == MIR_METRIC_EUCLIDEAN_DIST)) ? a.doc_sq[row] : 0.f;
#pragma unroll
            for (int i = 0; i < kXbQW; ++i) {
                if (!live[i]) continue;  // wave-uniform
                double dist;
                if (a.metric == MIR_METRIC_INNER_PRODUCT || a.metric == MIR_METRIC_COSINE_SIM) {
                    dist = -acc[i];
                } else {
                    const double sq = ((double)dsq - 2.0 * acc[i]) + qsq[i];
                    dist = a.metric == MIR_METRIC_SQEUCLIDEAN_DIST ? sq : sqrt(sq);
                }
                const double worst_d = __shfl(my_d[i], kk - 1, 64);
                const uint32_t worst_r = __shfl(my_r[i], kk - 1, 64);
                bool ok = row_ok && (cnt[i] < kk || dist_before(dist, row, worst_d, worst_r));
                if (bounded) {  // rounds after the first (k > 64): only rows strictly after the previous round's last result
                    const int qi = a.flagged[p0 + wave * kXbQW + i];
                    ok = ok && dist_before(a.bound_dist[qi], a.bound_row[qi], dist, row);
                }
                unsigned long long m = __ballot(ok);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const double nd = __shfl(dist, l, 64);
                    exact_wave_insert(nd, row0 + (uint32_t)l, kk, lane, my_d[i], my_r[i], cnt[i]);
                }
            }
        }
        // ---- this workgroup's lists of the pass -> HBM; the last workgroup to arrive merges a query's G lists
        if (wave_live) {
#pragma unroll
            for (int i = 0; i < kXbQW; ++i) {
                if (!live[i]) continue;
                const int f = p0 + wave * kXbQW + i;
                uint64_t *out = a.part + ((size_t)f * G + blockIdx.x) * a.list_stride * 2;
                if (lane < kk) {
                    out[2 * lane] = lane < cnt[i] ? (uint64_t)__double_as_longlong(my_d[i]) : 0;
                    out[2 * lane + 1] = lane < cnt[i] ? ((1ull << 32) | my_r[i]) : 0;
                }
            }
        }
        __threadfence();
        __syncthreads();
        if (tid < np) s_last[tid] = atomicAdd(&a.arrive[p0 + tid], 1u) == G - 1;
        __syncthreads();
        for (int c = 0; c < np; ++c) {
            if (!s_last[c]) continue;  // uniform per workgroup
            __threadfence();
            const int f = p0 + c;
            const int qi = a.flagged[f];
            double md = 0.0;
            uint32_t mr = 0;
            int mc = 0;
            for (uint32_t g = wave; g < G; g += kXbWaves) {
                const uint64_t *l = a.part + ((size_t)f * G + g) * a.list_stride * 2;
                for (int j = 0; j < kk; ++j) {
                    const uint64_t w1 = __hip_atomic_load(l + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(w1 >> 32)) break;
                    const double dj = __longlong_as_double((long long)__hip_atomic_load(l + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const double worst_d = __shfl(md, kk - 1, 64);
                    const uint32_t worst_r = __shfl(mr, kk - 1, 64);
                    if (mc == kk && !dist_before(dj, (uint32_t)w1, worst_d, worst_r)) break;  // the rest of this list is worse still
                    exact_wave_insert(dj, (uint32_t)w1, kk, lane, md, mr, mc);
                }
            }
            int total;
            const int rank = xb_block_rank<kXbWaves>(md, mr, mc, s_d, s_r, s_cnt, tid, &total);
            const int kout = total < kk ? total : kk;
            if (rank >= 0 && rank < kout) {
                const size_t o = (size_t)qi * a.k + (size_t)kExactRound * a.round + rank;
                if (a.out_row) a.out_row[o] = a.row_offset + (int64_t)mr;
                if (a.out_dist) a.out_dist[o] = md;
                if (a.out_doc) a.out_doc[o] = a.doc_ids ? a.doc_ids[mr] : 0;
                if (a.out_chunk) a.out_chunk[o] = a.chunk_ids ? a.chunk_ids[mr] : (int64_t)mr;
                if (rank == kout - 1) {
                    a.bound_dist[qi] = md;
                    a.bound_row[qi] = mr;
                }
            }
            if (tid == 0) {
                a.arrive[f] = 0;
                if (a.round == 0) {
                    if (a.out_count) a.out_count[qi] = (int)((uint32_t)a.k < a.n_rows ? (uint32_t)a.k : a.n_rows);
                    if (a.out_flags) a.out_flags[qi] = MIR_FLAG_EXACT_PASS;
                }
            }
            __syncthreads();
        }
    }
}

// One dispatch behind every search: nothing flagged -> exits at once; one or two flagged queries -> round 2's serial
// pass (a wave per row: 8.4 ms per query at 10M x 384, where the batched pass's sweep costs ~15 ms whatever the count:
// its scalar query loads are latency-bound with a single wave at work); more -> the batched pass.
constexpr int kXbSerialMax = 2;
static_assert(kXbThreads == kExactThreads, "one block size for both passes");
template <typename T, bool COS>
__global__ __launch_bounds__(kXbThreads) void exact_pass_kernel(ExactBatchArgs b, ExactArgs s) {
    const int nf = *b.nflag;
    if (nf == 0) return;
    if (nf <= kXbSerialMax) exact_topk_serial(s, nf);
    else exact_topk_batch<T, COS>(b, nf);
}

}  // namespace mir
