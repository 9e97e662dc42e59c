// Device code of the vector-search path (gfx950 / CDNA4 only).
//
// Replaces, for a batch of queries, the per-query numpy passes of
//   aidial_rag/retrievers/embeddings_metrics.py:14-50  (metric over [N,d])
//   aidial_rag/retrievers/embeddings_index.py:51-89    (stable argsort, first k)
//
// Data layout in HBM (DESIGN.md "Vector index layout"):
//   orig   f32 [N][d] row-major            - exact values, gathered only for the
//                                            few candidates that are re-scored
//   split  bf16 fragment-major             - what the scan streams.  Rows are
//          grouped in tiles of 32; a tile holds KS k-steps of 16 columns; each
//          k-step holds a `hi` and a `lo` block of 1 KiB = 64 lanes x 8 bf16,
//          lane l = (row l&31, column half l>>5) - exactly the A operand of
//          v_mfma_f32_32x32x16_bf16, so one coalesced 16-B-per-lane load IS the
//          fragment.  hi = bf16(x), lo = bf16(x - hi): x = hi + lo to 2^-18.
//   aux    f32 doc_sq[N] (numpy pairwise order, bit-exact), inv_norm[N]
//
// Scan (one pass of the shard per group of 32 queries): each wave owns whole
// tiles; the 32 queries' fragments live in 2*4*KS VGPRs for the whole kernel;
// per k-step three MFMAs (hi*hi, hi*lo, lo*hi) give the dot products to
// ~1.5e-5*|d||q| worst case.  With docs as A and queries as B, the 32x32
// accumulator puts ONE query on each lane (column = lane&31), so top-k
// selection is lane-local: a threshold compare per score and a rare insertion
// into that lane's list in LDS.  No barriers inside the pass.
//
// Finalize: per query, the per-workgroup lists are merged to the best `klist`
// candidates, those rows are re-scored in float64 with the reference's own
// formulas, ordered by (distance, row) - the reference's stable tie-break -
// and an a-posteriori bound check proves the candidate set was complete.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace mir {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kTileRows = 32;
constexpr int kMaxList = 64;         // per-lane candidate list cap (LDS: 64*256*8 = 128 KiB)
constexpr int kListMargin = 8;       // klist = k + margin
constexpr int kSortN = 8192;         // finalize: LDS bitonic width (64 KiB)
// bound on |scan dot - exact dot| / (|d||q|) of the three-product scans (hi.hi + hi.lo + lo.hi of x = hi + lo + r, |lo| <= u |x|,
// |r| <= u^2 |x|, u = 2^-8): the missing lo.lo and the two residuals are 3 u^2 = 4.6e-5, plus the float32 accumulation of the
// K products, K * 2^-24 of sum |x_i q_i| at worst (scan_rel_err()).  (Until late in round 3: 2e-5 flat, from u = 2^-9.)
constexpr double kScanSplitErr = 4.6e-5;
__host__ __device__ constexpr double scan_rel_err(int d) { return kScanSplitErr + 6.0e-8 * (d < 384 ? 384 : d); }

enum ScanKind { SCAN_IP = 0, SCAN_L2 = 1, SCAN_COS = 2 };

// ---------------------------------------------------------------- helpers

__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;  // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }

// x -> (hi, lo) bf16 bit patterns with x ~= hi + lo
__device__ __forceinline__ void split_bf16(float x, uint32_t &hi, uint32_t &lo) {
    hi = bf16_rne_bits(x);
    float r = x - bf16_bits_to_float(hi);  // exact in f32
    lo = bf16_rne_bits(r);
}

__device__ __forceinline__ uint4 pack8(const uint32_t (&b)[8]) {
    return make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
}

// Monotone float -> uint32 (larger float = larger uint).
__device__ __forceinline__ uint32_t orderable(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// Candidate key: larger is better; among equal values the LOWER row wins
// (embeddings_index.py:57-58 stable argsort).  0 is the "empty" sentinel.
__device__ __forceinline__ uint64_t make_key(float v, uint32_t row) {
    return ((uint64_t)orderable(v) << 32) | (uint64_t)(0xffffffffu - row);
}
__device__ __forceinline__ uint32_t key_row(uint64_t key) { return 0xffffffffu - (uint32_t)key; }
__device__ __forceinline__ float key_value(uint64_t key) { return unorderable((uint32_t)(key >> 32)); }

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
// the same butterfly over aligned groups of W lanes (W = 64: wave_sum)
template <int W>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int off = W / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// ---------------------------------------------------------------- index build

// f32 [n][d] row-major -> split fragment-major blocks.  One thread per
// (tile, k-step, lane).  ksteps*16 >= d; columns past d and rows past n are 0.
__global__ __launch_bounds__(256) void pack_split_f32_kernel(const float *__restrict__ src, int64_t n, int d,
                                                             int ksteps, int64_t total_lanes,
                                                             uint4 *__restrict__ dst) {
    int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_lanes) return;
    int lane = (int)(gid & 63);
    int64_t blk = gid >> 6;
    int s = (int)(blk % ksteps);
    int64_t tile = blk / ksteps;
    int64_t row = tile * kTileRows + (lane & 31);
    int col0 = 16 * s + 8 * (lane >> 5);
    float x[8];
    if (row < n && col0 + 8 <= d && (d & 3) == 0) {
        const float4 *p = reinterpret_cast<const float4 *>(src + row * (int64_t)d + col0);
        float4 a = p[0], b = p[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (row < n && col0 + j < d) ? src[row * (int64_t)d + col0 + j] : 0.f;
    }
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf16(x[j], hi[j], lo[j]);
    dst[(blk * 2 + 0) * 64 + lane] = pack8(hi);
    dst[(blk * 2 + 1) * 64 + lane] = pack8(lo);
}

// float16 [n*d] -> float32 (exact); grid-stride
__global__ __launch_bounds__(256) void widen_f16_kernel(const _Float16 *__restrict__ src, int64_t total,
                                                        float *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) dst[i] = (float)src[i];
}

// numpy's float32 pairwise sum of squares (numpy/_core/src/umath/loops_utils.h.src
// `@TYPE@_pairwise_sum`, block size 128, 8 accumulators) so that doc_sq matches
// `np.sum(docs**2, axis=1)` of embeddings_metrics.py:40 bit for bit.  The library
// is built with -ffp-contract=off (HIP's *_rn intrinsics are plain operators and
// would otherwise be fused into fma, which rounds differently); the
// _rn intrinsics keep the compiler from contracting mul+add into an fma.
__device__ __forceinline__ float sq_rn(float x) { return __fmul_rn(x, x); }

template <typename T>
__device__ inline float np_pairwise_leaf_sq(const T *a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = __fadd_rn(r, sq_rn((float)a[i]));
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = sq_rn((float)a[j]);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], sq_rn((float)a[i + j]));
    }
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __fadd_rn(res, sq_rn((float)a[i]));
    return res;
}
template <int DEPTH, typename T>
__device__ inline float np_pairwise_sq(const T *a, int n) {
    if (n <= 128) return np_pairwise_leaf_sq(a, n);
    if constexpr (DEPTH == 0) {
        // d > 128 * 2^12 is rejected on the host; unreachable
        return np_pairwise_leaf_sq(a, 128);
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return __fadd_rn(np_pairwise_sq<DEPTH - 1>(a, n2), np_pairwise_sq<DEPTH - 1>(a + n2, n - n2));
    }
}

// sum of squares in float64, strictly left to right; the loads of 16 elements are issued together (written as
// one load per add, hipcc waits for each: ~150 cycles per element, 5.4 ms per 2.5M x 384 rows)
template <typename T>
__device__ __forceinline__ double seq_sum_sq_f64(const T *a, int d) {
    double s = 0.0;
    int j = 0;
    for (; j + 16 <= d; j += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (float)a[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += (double)v[u] * (double)v[u];
    }
    for (; j < d; ++j) s += (double)a[j] * (double)a[j];
    return s;
}

// |x - bf16(x)|_2 of a row, rounded up: what the hi*hi filter of a float32 index loses on the row's side (Cauchy-Schwarz:
// |(x - hi) . q| <= that * |q|).  The norm kernels keep its maximum over the rows (word 2 of the norm statistics) and the
// maximum of its product with the row's inverse norm (word 3, the cosine form).
template <typename T>
__device__ __forceinline__ float bf16_residual_norm(const T *a, int d) {
    double s = 0.0;
    for (int j = 0; j < d; ++j) {
        const float x = (float)a[j];
        const float r = x - bf16_bits_to_float(bf16_rne_bits(x));  // exact in float32
        s += (double)r * (double)r;
    }
    return (float)sqrt(s) * (1.0f + 1e-6f);
}

// One thread per row: doc_sq (f32, numpy order), inv_norm = 1/max(|d|, 1e-8),
// and the running maximum row norm (for the scan's error bound).
// T = float, or _Float16 (a float16 index: the reference up-casts to float32 first, so the values
// summed are the same)
template <typename T>
__global__ __launch_bounds__(256) void row_norms_kernel(const T *__restrict__ src, int64_t n, int d,
                                                        float *__restrict__ doc_sq,
                                                        float *__restrict__ inv_norm,
                                                        unsigned int *__restrict__ max_norm_bits) {
    int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float nrm = 0.f, res = 0.f, rel = 0.f;
    if (row < n) {
        const T *a = src + row * (int64_t)d;
        doc_sq[row] = np_pairwise_sq<12>(a, d);
        const double s = seq_sum_sq_f64(a, d);
        nrm = (float)sqrt(s);
        nrm = nrm * (1.0f + 1e-6f);  // round up: used as an upper bound
        inv_norm[row] = 1.0f / fmaxf((float)sqrt(s), 1e-8f);
        if (sizeof(T) == 4) {
            res = bf16_residual_norm(a, d);
            rel = res * inv_norm[row] * (1.0f + 1e-6f);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { res = fmaxf(res, __shfl_xor(res, off, 64)); rel = fmaxf(rel, __shfl_xor(rel, off, 64)); }
    if ((threadIdx.x & 63) == 0 && res > 0.f) { atomicMax(max_norm_bits + 2, __float_as_uint(res)); atomicMax(max_norm_bits + 3, __float_as_uint(rel)); }
    // wave max, then one atomic per wave (positive floats order as uints)
    // word 1: some row holds a NaN or an infinity (fmaxf drops a NaN: the maximum alone would not tell)
    if (__ballot(!(nrm < __builtin_inff())) != 0ull && (threadIdx.x & 63) == 0) atomicOr(max_norm_bits + 1, 1u);
    for (int off = 32; off >= 1; off >>= 1) nrm = fmaxf(nrm, __shfl_xor(nrm, off, 64));
    if ((threadIdx.x & 63) == 0 && nrm > 0.f) atomicMax(max_norm_bits, __float_as_uint(nrm));
}

// The same per-row arithmetic with the rows staged through LDS: a workgroup copies `rows_per_wg` consecutive rows
// (one contiguous piece of HBM, read coalesced by all 256 threads), then thread r sums row r out of LDS (row
// stride odd in banks: no conflicts).  One thread per row straight from HBM reads every row as its own stream of
// cache lines and reaches ~0.7 TB/s; this form is bound by the copy.  LDS holds float32 whatever T is (the
// float16 -> float32 conversion is exact and is what the direct kernel feeds its sums too).
constexpr int kNormsLdsBytes = 64 * 1024;  // two workgroups per CU
__host__ __device__ inline int norms_row_stride(int d) { return d | 1; }
template <typename T>
__global__ __launch_bounds__(256) void row_norms_lds_kernel(const T *__restrict__ src, int64_t n, int d, int rows_per_wg,
                                                            float *__restrict__ doc_sq, float *__restrict__ inv_norm,
                                                            unsigned int *__restrict__ max_norm_bits) {
    extern __shared__ float norms_lds[];
    const int stride = norms_row_stride(d);
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg;
    const int rows = (int)(n - row0 < rows_per_wg ? n - row0 : rows_per_wg);
    const T *base = src + row0 * (int64_t)d;
    // the rows are one contiguous piece of `total` elements; 16 loads in flight per thread (a load per LDS store
    // is waited for one by one: 84 exposed HBM round trips per workgroup)
    const int total = rows * d;
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 16) {
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + 256 * u;
            v[u] = e < total ? base[e] : T(0);
        }
        int r = e0 / d, c = e0 - r * d;  // one division per batch, then (row, column) advance by 256 elements
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (e0 + 256 * u < total) norms_lds[r * stride + c] = (float)v[u];
            c += 256;
            while (c >= d) { c -= d; ++r; }
        }
    }
    __syncthreads();
    // the two sums of a row are independent: threads [0, rows) of the first half of the workgroup take the float32
    // pairwise sum, threads [128, 128 + rows) the float64 one (rows <= 128 by construction of rows_per_wg)
    float nrm = 0.f, res = 0.f, rel = 0.f;
    const int rt = threadIdx.x & 127;
    if (rt < rows) {
        const float *a = norms_lds + rt * stride;
        const int64_t row = row0 + rt;
        if (threadIdx.x < 128) {
            doc_sq[row] = np_pairwise_sq<12>(a, d);
        } else {
            const double s = seq_sum_sq_f64(a, d);
            nrm = (float)sqrt(s);
            nrm = nrm * (1.0f + 1e-6f);
            inv_norm[row] = 1.0f / fmaxf((float)sqrt(s), 1e-8f);
            if (sizeof(T) == 4) {
                res = bf16_residual_norm(a, d);
                rel = res * inv_norm[row] * (1.0f + 1e-6f);
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { res = fmaxf(res, __shfl_xor(res, off, 64)); rel = fmaxf(rel, __shfl_xor(rel, off, 64)); }
    if ((threadIdx.x & 63) == 0 && res > 0.f) { atomicMax(max_norm_bits + 2, __float_as_uint(res)); atomicMax(max_norm_bits + 3, __float_as_uint(rel)); }
    // word 1: some row holds a NaN or an infinity (fmaxf drops a NaN: the maximum alone would not tell)
    if (__ballot(!(nrm < __builtin_inff())) != 0ull && (threadIdx.x & 63) == 0) atomicOr(max_norm_bits + 1, 1u);
    for (int off = 32; off >= 1; off >>= 1) nrm = fmaxf(nrm, __shfl_xor(nrm, off, 64));
    if ((threadIdx.x & 63) == 0 && nrm > 0.f) atomicMax(max_norm_bits, __float_as_uint(nrm));
}

// ---------------------------------------------------------------- query prep

// Blocks [0, ngroups*ksteps): write the B-operand fragments of query group g,
// k-step s (lane l = query 32g + (l&31), columns 16s + 8(l>>5) ..+7).
// Blocks [ngroups*ksteps, +b): per-query sum of squares and norm in float64.
__global__ __launch_bounds__(64) void prep_queries_kernel(const double *__restrict__ q, int b, int d, int ksteps,
                                                          int ngroups, uint4 *__restrict__ qsplit,
                                                          double *__restrict__ q_sq,
                                                          double *__restrict__ q_norm,
                                                          unsigned long long *__restrict__ gthr, int gthr_words) {
    int lane = threadIdx.x;
    int blk = blockIdx.x;
    // starting thresholds of the wide scans: zero = none (was a separate memset node, ~10 us of dependent launch)
    if (gthr && blk * 64 + lane < gthr_words) gthr[blk * 64 + lane] = 0;
    if (blk < ngroups * ksteps) {
        int s = blk % ksteps;
        int g = blk / ksteps;
        int qi = 32 * g + (lane & 31);
        int col0 = 16 * s + 8 * (lane >> 5);
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = (qi < b && col0 + j < d) ? (float)q[(int64_t)qi * d + col0 + j] : 0.f;
            split_bf16(x, hi[j], lo[j]);
        }
        qsplit[((int64_t)blk * 2 + 0) * 64 + lane] = pack8(hi);
        qsplit[((int64_t)blk * 2 + 1) * 64 + lane] = pack8(lo);
    } else {
        int qi = blk - ngroups * ksteps;
        if (qi >= b) return;
        double s = 0.0;
        for (int j = lane; j < d; j += 64) {
            double x = q[(int64_t)qi * d + j];
            s += x * x;
        }
        s = wave_sum(s);
        if (lane == 0) {
            q_sq[qi] = s;
            q_norm[qi] = sqrt(s);
        }
    }
}

// ---------------------------------------------------------------- scan

// Replace this lane's current worst entry, then find the new worst.  The rescan
// reads 8 entries at a time before comparing, so the LDS latency is paid once
// per 8 entries instead of once per entry (klist is a multiple of 2, padded
// reads past klist see the ~0 sentinel rows of the caller's allocation: no -
// they are masked by the bound check below).
template <int STRIDE = 256>
__device__ __forceinline__ void list_insert(uint64_t *list, int klist, int tid, uint64_t key, uint64_t &minkey,
                                            int &minpos) {
    list[minpos * STRIDE + tid] = key;
    uint64_t m = ~0ull;
    int mp = 0;
    for (int p0 = 0; p0 < klist; p0 += 8) {
        uint64_t x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (p0 + j < klist) ? list[(p0 + j) * STRIDE + tid] : ~0ull;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (x[j] < m) {
                m = x[j];
                mp = p0 + j;
            }
        }
    }
    minkey = m;
    minpos = mp;
}

// Epilogue of one 32x32 tile for this lane's query: 16 scores (C layout: row =
// (r&3) + 8*(r>>2) + 4*(lane>>5)).  A 16-bit mask marks the scores that beat the
// lane's current worst entry; each lane then walks ITS OWN set bits, so the
// wave runs max-over-lanes(popcount) insertions instead of one (exec-masked)
// insertion per register position that any lane needs, and the insertion code
// exists once instead of 16 times.
template <int KIND>
__device__ __forceinline__ void tile_epilogue(const f32x16 &acc_m, const f32x16 &acc_c, const float4 (&ax)[4],
                                              uint32_t row0, uint32_t n_rows, uint64_t *list, int klist, int tid,
                                              uint64_t &minkey, int &minpos) {
    float v[16];
    uint32_t mask = 0;
    const float vmin = key_value(minkey);  // minkey == 0 decodes to NaN: compare below is then false -> use flag
    const bool open = minkey == 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = r >> 2, i = r & 3;
        const float dot = acc_m[r] + acc_c[r];
        float x;
        if (KIND == SCAN_IP) {
            x = dot;
        } else {
            const float a = (i == 0) ? ax[g].x : (i == 1) ? ax[g].y : (i == 2) ? ax[g].z : ax[g].w;
            x = (KIND == SCAN_L2) ? fmaf(2.0f, dot, -a) : dot * a;
        }
        x = (x == x) ? x + 0.0f : -__builtin_inff();  // NaN ranks last; -0 -> +0
        v[r] = x;
        const uint32_t row = row0 + 8 * g + i;
        if (row < n_rows && (open || x >= vmin)) mask |= 1u << r;
    }
    while (mask) {
        const int r = __builtin_ctz(mask);
        mask &= mask - 1;
        float x = v[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) x = (r == j) ? v[j] : x;
        const uint32_t row = row0 + 8 * (r >> 2) + (r & 3);
        const uint64_t key = make_key(x, row);
        if (key > minkey) list_insert(list, klist, tid, key, minkey, minpos);
    }
}

// grid = (#CUs), block = 256 (4 waves, one per SIMD, whole register file each).
// dynamic LDS = (klist*256 + 32*klist) * 8 bytes.
template <int KSTEPS, int KIND>
__global__ __launch_bounds__(256, 1) void scan_topk_kernel(const uint4 *__restrict__ docs,
                                                           const float *__restrict__ aux,
                                                           const uint4 *__restrict__ qsplit, uint32_t n_rows,
                                                           uint32_t n_tiles, int nq, int klist,
                                                           uint64_t *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *list = reinterpret_cast<uint64_t *>(smem);  // [klist][256], lane-private columns
    uint64_t *stage = list + klist * 256;                 // [32][klist]

    constexpr int R = KSTEPS < 8 ? KSTEPS : 8;  // ring depth in k-steps (2 KiB each)
    static_assert(KSTEPS % R == 0, "ring slot must be static across tiles");
    constexpr int TILE_U4 = KSTEPS * 128;  // uint4 per tile (hi+lo)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int qj = lane & 31;
    const uint32_t wave_global = blockIdx.x * 4 + (tid >> 6);
    const uint32_t total_waves = gridDim.x * 4;

    for (int p = 0; p < klist; ++p) list[p * 256 + tid] = 0;
    uint64_t minkey = 0;
    int minpos = 0;

    if (wave_global < n_tiles) {
        // the 32 queries of this group, as B-operand fragments, for the whole kernel
        bf16x8 qh[KSTEPS], ql[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            qh[s] = __builtin_bit_cast(bf16x8, qsplit[(s * 2 + 0) * 64 + lane]);
            ql[s] = __builtin_bit_cast(bf16x8, qsplit[(s * 2 + 1) * 64 + lane]);
        }

        uint32_t t = wave_global;
        const uint4 *tp = docs + (size_t)t * TILE_U4 + lane;
        uint4 bh[R], bl[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            bh[i] = tp[(2 * i + 0) * 64];
            bl[i] = tp[(2 * i + 1) * 64];
        }

        while (true) {
            const uint32_t tn = t + total_waves;
            const bool more = tn < n_tiles;
            // prefetch target for the ring's wrap-around: next tile, or this one again at the end
            const uint4 *np = docs + (size_t)(more ? tn : t) * TILE_U4 + lane;

            float4 ax[4];
            if (KIND != SCAN_IP) {
                const float4 *ap = reinterpret_cast<const float4 *>(aux + (size_t)t * kTileRows);
#pragma unroll
                for (int g = 0; g < 4; ++g) ax[g] = ap[2 * g + h];
            }

            f32x16 acc_m = {0}, acc_c = {0};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int slot = ks % R;
                bf16x8 ah = __builtin_bit_cast(bf16x8, bh[slot]);
                bf16x8 al = __builtin_bit_cast(bf16x8, bl[slot]);
                acc_m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[ks], acc_m, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[ks], acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[ks], acc_c, 0, 0, 0);
                if (ks + R < KSTEPS) {
                    bh[slot] = tp[(2 * (ks + R) + 0) * 64];
                    bl[slot] = tp[(2 * (ks + R) + 1) * 64];
                } else {
                    bh[slot] = np[(2 * (ks + R - KSTEPS) + 0) * 64];
                    bl[slot] = np[(2 * (ks + R - KSTEPS) + 1) * 64];
                }
                // Pin the k-step: without this hipcc sinks each refill next to its
                // use R steps later (fewer live registers) and the wave ends up with
                // ~2 loads in flight instead of 2*R.
                __builtin_amdgcn_sched_barrier(0);
            }

            if (qj < nq) tile_epilogue<KIND>(acc_m, acc_c, ax, t * kTileRows + 4 * h, n_rows, list, klist, tid, minkey, minpos);
            if (!more) break;
            t = tn;
            tp = np;
        }
    }

    // ---- in-workgroup merge: 8 lane lists per query -> one list of klist ----
    __syncthreads();
    for (int i = tid; i < 32 * klist; i += 256) stage[i] = 0;
    __syncthreads();
    for (int p = 0; p < klist; ++p) {
        const uint64_t key = list[p * 256 + tid];
        if (key == 0) continue;
        int rank = 0;
        for (int l = 0; l < 8; ++l) {
            const int t2 = 64 * (l >> 1) + 32 * (l & 1) + qj;
            for (int p2 = 0; p2 < klist; ++p2) rank += (list[p2 * 256 + t2] > key) ? 1 : 0;
        }
        if (rank < klist) stage[qj * klist + rank] = key;  // keys are distinct: ranks are too
    }
    __syncthreads();
    uint64_t *out = part + (size_t)blockIdx.x * 32 * klist;
    for (int i = tid; i < 32 * klist; i += 256) out[i] = stage[i];
}

// ---------------------------------------------------------------- shared pieces of the wide (64- and 128-query) scans
// The doc stream costs the same HBM bytes whether 32 or 128 queries ride on it, so the wide scans (vec_kernels_q16.h:
// 128 queries over the float32 image of d <= 384; vec_kernels_f16.h: 64 queries over float16 / wide float32 rows)
// bring it into LDS ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4; the images are lane-linear, so a 1-KiB
// piece is one wave-instruction and ds_read_b128 at lane*16 is conflict-free) and every wave reads it from there: a
// ring of stages, per stage ONE raw s_barrier and a COUNTED vmcnt (never 0 in steady state) so that several stages stay
// in flight across barriers.  No VGPR-destination global load exists in their loops (hipcc would drain vmcnt(0) for
// it): the per-row norm column comes through the scalar cache, it is wave-uniform.
constexpr int kB128Pending = 4;  // per-lane buffer of appended, not yet merged candidates (vec_kernels_f16.h)

// LDS-DMA of one 1-KiB piece (16 B per lane) as inline asm: with the builtin, hipcc
// orders every later ds_read behind ALL pending DMAs (`s_waitcnt vmcnt(0)` after the
// first ds_read of each stage), which drains the ring; an asm load is outside its
// bookkeeping, so only the counted waits below apply.  M0 carries the wave-uniform
// LDS byte address and is saved/restored inside the statement (it is compiler-owned).
__device__ __forceinline__ void glds16_b128(const void *gsrc, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}
// the same for 4 bytes per (active) lane: lane l's dword lands at lds_byte_addr + 4 l
__device__ __forceinline__ void glds4_b32(const void *gsrc, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}
// the same with a wave-uniform base (an SGPR pair) and a 32-bit byte offset per lane: no 64-bit address to keep in VGPRs
__device__ __forceinline__ void glds4_b32_sv(const void *gbase_uniform, uint32_t lane_byte_off, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_byte_addr)
        : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// Candidate handling of the 64-query K-split scan (vec_kernels_f16.h).  Its waves run in lockstep (one
// barrier per stage), so anything lane-divergent with a dependent LDS chain sits
// on the workgroup's critical path at almost every tile (measured: insertions
// cost 2.5 ms of a 5.6 ms launch).  Therefore:
//  * a passing score is only APPENDED to the lane's pending buffer (one LDS
//    write); when any lane's buffer is full the whole wave merges its buffers
//    into the lists at once.  The threshold is the list's worst entry as of the
//    last merge (slightly stale = a few more appends, never a wrong result);
//  * the per-score filter of tile t-1 is software-pipelined into tile t: one
//    score after each MFMA of the first stage, so its VALU work issues in the
//    matrix pipe's shadow instead of extending the tile (see the kernel).
template <int STRIDE = 256>
__device__ __forceinline__ void drain_candidates(uint32_t mask, const float (&v)[16], uint32_t row0, uint64_t *list,
                                                 int klist, int tid, uint64_t &minkey, int &minpos, int &pending) {
    while (__any(mask != 0)) {
        if (mask) {
            const int r = __builtin_ctz(mask);
            mask &= mask - 1;
            // v[r] for a lane-dependent r, kept in registers: left as a plain select chain, hipcc
            // rewrites it into an indexed scratch array + `s_waitcnt vmcnt(0)`, and that wait drains
            // the whole DMA ring at every candidate.  The empty asm makes each element opaque.
            float x = v[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                float c = v[j];
                asm volatile("" : "+v"(c));
                x = (r == j) ? c : x;
            }
            x = (x == x) ? x + 0.0f : -__builtin_inff();  // NaN ranks last; -0 -> +0
            const uint64_t key = make_key(x, row0 + 8 * (r >> 2) + (r & 3));
            if (key > minkey) {
                list[(klist + pending) * STRIDE + tid] = key;
                ++pending;
            }
        }
        if (__any(pending == kB128Pending)) {  // wave-uniform: merge every lane's buffer now
            for (int i = 0; i < kB128Pending; ++i) {
                if (i < pending) {
                    const uint64_t key = list[(klist + i) * STRIDE + tid];
                    if (key > minkey) list_insert<STRIDE>(list, klist, tid, key, minkey, minpos);
                }
            }
            pending = 0;
        }
    }
}

// Generic dimension: query fragments are re-read from L2 each k-step instead of
// living in registers.  Same tile walk, same epilogue.  ksteps % 8 == 0.
template <int KIND>
__global__ __launch_bounds__(256, 1) void scan_topk_generic_kernel(const uint4 *__restrict__ docs,
                                                                   const float *__restrict__ aux,
                                                                   const uint4 *__restrict__ qsplit,
                                                                   int ksteps, uint32_t n_rows,
                                                                   uint32_t n_tiles, int nq, int klist,
                                                                   uint64_t *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *list = reinterpret_cast<uint64_t *>(smem);
    uint64_t *stage = list + klist * 256;
    constexpr int R = 8;
    const size_t tile_u4 = (size_t)ksteps * 128;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, qj = lane & 31;
    const uint32_t wave_global = blockIdx.x * 4 + (tid >> 6);
    const uint32_t total_waves = gridDim.x * 4;
    for (int p = 0; p < klist; ++p) list[p * 256 + tid] = 0;
    uint64_t minkey = 0;
    int minpos = 0;

    for (uint32_t t = wave_global; t < n_tiles; t += total_waves) {
        const uint4 *tp = docs + (size_t)t * tile_u4 + lane;
        const uint4 *qp = qsplit + lane;
        float4 ax[4];
        if (KIND != SCAN_IP) {
            const float4 *ap = reinterpret_cast<const float4 *>(aux + (size_t)t * kTileRows);
#pragma unroll
            for (int g = 0; g < 4; ++g) ax[g] = ap[2 * g + h];
        }
        f32x16 acc_m = {0}, acc_c = {0};
        for (int k0 = 0; k0 < ksteps; k0 += R) {
            uint4 dh[R], dl[R], qh[R], ql[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                dh[i] = tp[(2 * (k0 + i) + 0) * 64];
                dl[i] = tp[(2 * (k0 + i) + 1) * 64];
                qh[i] = qp[(2 * (k0 + i) + 0) * 64];
                ql[i] = qp[(2 * (k0 + i) + 1) * 64];
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                bf16x8 ah = __builtin_bit_cast(bf16x8, dh[i]), al = __builtin_bit_cast(bf16x8, dl[i]);
                bf16x8 bh = __builtin_bit_cast(bf16x8, qh[i]), bl = __builtin_bit_cast(bf16x8, ql[i]);
                acc_m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc_m, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc_c, 0, 0, 0);
            }
        }
        if (qj < nq) tile_epilogue<KIND>(acc_m, acc_c, ax, t * kTileRows + 4 * h, n_rows, list, klist, tid, minkey, minpos);
    }
    __syncthreads();
    for (int i = tid; i < 32 * klist; i += 256) stage[i] = 0;
    __syncthreads();
    for (int p = 0; p < klist; ++p) {
        const uint64_t key = list[p * 256 + tid];
        if (key == 0) continue;
        int rank = 0;
        for (int l = 0; l < 8; ++l) {
            const int t2 = 64 * (l >> 1) + 32 * (l & 1) + qj;
            for (int p2 = 0; p2 < klist; ++p2) rank += (list[p2 * 256 + t2] > key) ? 1 : 0;
        }
        if (rank < klist) stage[qj * klist + rank] = key;
    }
    __syncthreads();
    uint64_t *out = part + (size_t)blockIdx.x * 32 * klist;
    for (int i = tid; i < 32 * klist; i += 256) out[i] = stage[i];
}

// ---------------------------------------------------------------- exact metric

// The reference's arithmetic for ONE (query, row) pair in float64, computed
// by a whole wave; every lane returns the same value.
//   inner_product    embeddings_metrics.py:20     -dot
//   sqeuclidean_dist embeddings_metrics.py:40-43  doc_sq(f32, numpy order) - 2*dot + q_sq
//   euclidean_dist   embeddings_metrics.py:50     sqrt of that (NaN if negative, as upstream)
//   cosine_sim       embeddings_metrics.py:28-31  torch: unit doc row rounded to f32
//                                                 (norm and division in f32), query unit in f64
// `rank_value` returns the quantity the scan ranks by, in the scan's units.
template <typename T, int W = 64, int U = 8>
__device__ __forceinline__ double exact_metric_wave(const T *__restrict__ row, const double *__restrict__ q,
                                                    int d, int metric, float doc_sq32, double q_sq,
                                                    double q_norm, int lane, double *rank_value) {
    // W lanes (a whole wave, or an aligned group of 16: `lane` = the lane's index in its group, groups work on different rows)
    // share a row: lane l takes elements l, l + W, ... in that order.  Eight of them are loaded before the first is used:
    // written load-by-use, every element of a (cold) row was its own exposed HBM round trip.  (U = loads in flight per lane.)
    if (metric == MIR_METRIC_COSINE_SIM) {
        double s = 0.0;
        for (int j0 = lane; j0 < d; j0 += W * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = j0 + W * u < d ? (float)row[j0 + W * u] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + W * u < d) {
                    const double x = (double)v[u];
                    s += x * x;
                }
        }
        s = group_sum<W>(s);
        const float dn = fmaxf((float)sqrt(s), 1e-8f);
        const double qn = fmax(q_norm, 1e-8);
        double c = 0.0;
        for (int j0 = lane; j0 < d; j0 += W * U) {
            float v[U];
            double qv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = j0 + W * u < d;
                v[u] = in ? (float)row[j0 + W * u] : 0.f;
                qv[u] = in ? q[j0 + W * u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + W * u < d) c += (double)__fdiv_rn(v[u], dn) * (qv[u] / qn);
        }
        c = group_sum<W>(c);
        *rank_value = c * qn;
        return -c;
    }
    double dot = 0.0;
    for (int j0 = lane; j0 < d; j0 += W * U) {
        float v[U];
        double qv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = j0 + W * u < d;
            v[u] = in ? (float)row[j0 + W * u] : 0.f;
            qv[u] = in ? q[j0 + W * u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j0 + W * u < d) dot += (double)v[u] * qv[u];
    }
    dot = group_sum<W>(dot);
    if (metric == MIR_METRIC_INNER_PRODUCT) {
        *rank_value = dot;
        return -dot;
    }
    const double sq = ((double)doc_sq32 - 2.0 * dot) + q_sq;
    *rank_value = 2.0 * dot - (double)doc_sq32;
    return metric == MIR_METRIC_SQEUCLIDEAN_DIST ? sq : sqrt(sq);
}

// (a better than b) under the reference's ordering: distance ascending, NaN
// last (numpy sort order), then flattened row ascending (stable argsort).
__device__ __forceinline__ bool dist_before(double da, uint32_t ra, double db, uint32_t rb) {
    const bool na = da != da, nb = db != db;
    if (na || nb) return na == nb ? ra < rb : nb;
    return da < db || (da == db && ra < rb);
}

// One wave per row: out[row] = metric(query, docs[row]).  grid = ceil(n/4), block 256.
template <typename T>
__global__ __launch_bounds__(256) void metric_eval_kernel(const T *__restrict__ docs,
                                                          const float *__restrict__ doc_sq, int64_t n, int d,
                                                          const double *__restrict__ q,
                                                          const double *__restrict__ q_sq,
                                                          const double *__restrict__ q_norm, int metric,
                                                          double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    double rv;
    double dist = exact_metric_wave(docs + row * (int64_t)d, q, d, metric, doc_sq[row], q_sq[0], q_norm[0], lane, &rv);
    if (lane == 0) out[row] = dist;
}

// ---------------------------------------------------------------- finalize

// In-LDS bitonic sort, descending, n = power of two, 256 threads.
__device__ inline void bitonic_sort_desc(uint64_t *keys, int n, int tid) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (n >> 1); i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// Best `klist` keys of `nwg` per-workgroup lists that are each already sorted (the scan kernels
// emit their lists by rank).  A tournament over the list heads: every thread owns up to 4 lists,
// per round one block-wide max picks the winner and only its owner advances.  ~klist rounds of a
// few hundred cycles instead of a bitonic sort of all nwg*klist keys.  Keys are distinct (they
// carry the row), 0 = empty.  out[0..klist) receives the keys, best first, 0-padded.
__device__ inline void merge_sorted_lists(const uint64_t *__restrict__ lists, size_t list_stride, int nwg, int klist,
                                          uint64_t *out, uint64_t *red /*[4]*/, int tid) {
    // cur = head of each owned list; n1, n2 = the two entries behind it, fetched up front with it (one round
    // trip for all 12 loads).  A load issued inside the loop is waited for at the next barrier by everybody, a
    // round trip per round; with 256 lists a list rarely gives more than three of the best klist.
    int head[4];
    uint64_t cur[4], n1[4], n2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int wg = tid + 256 * j;
        const bool mine_j = wg < nwg;
        const uint64_t *l = lists + (size_t)wg * list_stride;
        head[j] = 0;
        cur[j] = mine_j ? l[0] : 0;
        n1[j] = (mine_j && klist > 1) ? l[1] : 0;
        n2[j] = (mine_j && klist > 2) ? l[2] : 0;
    }
    for (int r = 0; r < klist; ++r) {
        uint64_t mine = cur[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) mine = cur[j] > mine ? cur[j] : mine;
        uint64_t best = mine;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(best >> 32), off, 64) << 32) |
                               (uint64_t)__shfl_xor((uint32_t)best, off, 64);
            best = o > best ? o : best;
        }
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        best = red[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) best = red[w] > best ? red[w] : best;
        __syncthreads();
        if (tid == 0) out[r] = best;
        if (best != 0 && mine == best) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cur[j] == best) {
                    ++head[j];
                    if (head[j] == 1) cur[j] = n1[j];
                    else if (head[j] == 2) cur[j] = n2[j];
                    else cur[j] = head[j] < klist ? lists[(size_t)(tid + 256 * j) * list_stride + head[j]] : 0;
                }
            }
        }
    }
    __syncthreads();
}

// ---- hand-over to the exact pass (vec_kernels_exact.h): the flagging kernels write the query's column of the transposed
// copy Qt[pass][column][32 queries] that the batched pass reads through the scalar cache
constexpr int kXbCols = 192;     // columns staged per slice of the exact pass
constexpr int kXbQ = 128;        // queries per pass of the exact pass
__host__ __device__ constexpr int xb_dpad(int d) { return (d + kXbCols - 1) / kXbCols * kXbCols; }
// whole block (`nthreads` threads), f = the query's position in the flagged list; cosine_sim stores q / max(|q|, 1e-8)
__device__ __forceinline__ void exact_publish_query(double *__restrict__ qt, int f, const double *__restrict__ q, int d, int metric,
                                                    double q_norm, int tid, int nthreads) {
    const int dpad = xb_dpad(d);
    const double qn = fmax(q_norm, 1e-8);
    double *dst = qt + ((size_t)(f / kXbQ) * dpad) * kXbQ + (f % kXbQ);
    for (int j = tid; j < dpad; j += nthreads) {
        double x = j < d ? q[j] : 0.0;
        if (metric == MIR_METRIC_COSINE_SIM) x = x / qn;
        dst[(size_t)j * kXbQ] = x;
    }
}

struct FinalizeArgs {
    const uint64_t *part;   // [launch][nwg][qpw][klist]
    int nwg;                // workgroups of the scan
    int qpw;                // queries per scan launch (32 or 128)
    int klist;
    int k;
    int b;
    int d;
    int metric;
    const float *docs;      // f32 [n][d], or null with
    const _Float16 *docs16; // f16 [n][d] (float16-native index)
    const float *doc_sq;    // f32 [n]
    const float *max_norm;  // 1 float
    double scan_rel_err;    // bound on |list value's dot - exact dot| / (|d||q|): scan_rel_err(d), or the hi-only float16 scan's
    const double *q;        // [b][d]
    const double *q_sq;     // [b]
    const double *q_norm;   // [b]
    const int64_t *chunk_ids;  // [n] or null
    const int32_t *doc_ids;    // [n] or null
    int64_t row_offset;
    int32_t *out_doc;
    int64_t *out_chunk;
    int64_t *out_row;
    double *out_dist;
    int32_t *out_count;
    int32_t *out_flags;
    int32_t *nflag;         // number of queries handed to the exact pass (zeroed by the prep kernel)
    int32_t *flagged;       // [b] their indices, in arrival order
    double *qt;             // the exact pass's transposed copy of the flagged queries
};

// grid = b (one block per query), block = 256, static LDS 64 KiB + small.
// Where its ~27 us go (128 queries, 256 lists of 12): ~5 dispatch, ~10 the 12 tournament rounds (a 64-bit
// wave max through ds_bpermute + two barriers each), the rest dependent HBM hops (list heads, candidate rows,
// norms).  Re-scoring on 16 waves instead of 4 and prefetching the lists' next entries changed nothing measurable.
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a) {
    __shared__ uint64_t keys[kMaxList];
    __shared__ uint64_t red[4];
    __shared__ double c_dist[kMaxList];
    __shared__ double c_rank[kMaxList];
    __shared__ uint32_t c_row[kMaxList];
    __shared__ double s_vk;
    __shared__ int s_f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = blockIdx.x;
    const int g = qi / a.qpw, ql = qi % a.qpw;
    const int klist = a.klist;
    const uint64_t *pg = a.part + (size_t)g * a.nwg * a.qpw * klist;

    // ---- 1. best klist candidate keys over all workgroups (tournament over the sorted lists) ----
    merge_sorted_lists(pg + (size_t)ql * klist, (size_t)a.qpw * klist, a.nwg, klist, keys, red, tid);
    int nc = 0;
    for (int i = 0; i < klist; ++i) nc += keys[i] != 0 ? 1 : 0;  // valid ones first

    // ---- 2. re-score the candidates exactly (float64, reference formulas) ----
    const double *q = a.q + (size_t)qi * a.d;
    for (int c = wave; c < nc; c += 4) {
        const uint32_t row = key_row(keys[c]);
        double rv;
        const double dist = a.docs16 ? exact_metric_wave(a.docs16 + (size_t)row * a.d, q, a.d, a.metric, a.doc_sq[row],
                                                         a.q_sq[qi], a.q_norm[qi], lane, &rv)
                                     : exact_metric_wave(a.docs + (size_t)row * a.d, q, a.d, a.metric, a.doc_sq[row],
                                                         a.q_sq[qi], a.q_norm[qi], lane, &rv);
        if (lane == 0) {
            c_dist[c] = dist;
            c_rank[c] = rv;
            c_row[c] = row;
        }
    }
    __syncthreads();

    // ---- 3. order by (distance, row) and emit the first k ----
    const int kout = a.k < nc ? a.k : nc;
    if (tid < nc) {
        int rank = 0;
        for (int c = 0; c < nc; ++c)
            if (c != tid && dist_before(c_dist[c], c_row[c], c_dist[tid], c_row[tid])) ++rank;
        if (rank < a.k) {
            const size_t o = (size_t)qi * a.k + rank;
            const uint32_t row = c_row[tid];
            if (a.out_row) a.out_row[o] = a.row_offset + (int64_t)row;
            if (a.out_dist) a.out_dist[o] = c_dist[tid];
            if (a.out_doc) a.out_doc[o] = a.doc_ids ? a.doc_ids[row] : 0;
            if (a.out_chunk) a.out_chunk[o] = a.chunk_ids ? a.chunk_ids[row] : (int64_t)row;
        }
        if (rank == kout - 1) s_vk = c_rank[tid];
    }
    __syncthreads();

    // ---- 4. a-posteriori completeness check ----
    // Every row NOT among the candidates has scan value <= tau (the worst kept
    // key); its exact value is <= tau + eps.  If that is below the exact value
    // of the k-th result, no excluded row can belong to the top k.
    if (tid == 0) {
        if (a.out_count) a.out_count[qi] = kout;
        int flag = 0;
        if (nc == klist && kout > 0) {
            const double tau = (double)key_value(keys[klist - 1]);
            const double qn = a.q_norm[qi];
            const double mx = (double)a.max_norm[0];
            double eps = a.scan_rel_err * qn * (a.metric == MIR_METRIC_COSINE_SIM ? 1.0 : mx);
            if (a.metric == MIR_METRIC_SQEUCLIDEAN_DIST || a.metric == MIR_METRIC_EUCLIDEAN_DIST) eps *= 2.0;
            eps += 1e-6 * fabs(tau);
            const double vk = s_vk;
            if (!(tau + eps < vk)) flag = MIR_FLAG_UNCERTAIN;
        }
        // An unproven query is handed to exact_topk_kernel (enqueued right behind this kernel), which
        // overwrites its outputs and flag; the reference is always exact (embeddings_index.py:51-60).
        s_f = -1;
        if (flag) {
            s_f = atomicAdd(a.nflag, 1);
            a.flagged[s_f] = qi;
        }
        if (a.out_flags) a.out_flags[qi] = flag;
    }
    __syncthreads();
    if (s_f >= 0) exact_publish_query(a.qt, s_f, q, a.d, a.metric, a.q_norm[qi], tid, 256);
}

// ---------------------------------------------------------------- exact pass
// The filter scan + completeness check above PROVES its result for almost every query; the rest -
// more than klist - k rows inside the scan's error band at the cut (near-duplicate chunks, boiler-plate
// pages), exact ties across the cut - and every query whose k exceeds the scan's candidate lists
// take this pass: the reference's own computation (embeddings_index.py:51-60,62-89: the metric in
// float64 for EVERY row, then the stable order on (distance, row)), with no assumption on the data.
//
// grid = #CUs x 1024 threads; a wave walks rows wave, wave + #waves, ... in ascending order, computes
// the reference formula for each (exact_metric_wave) and keeps its best `kk` <= 64 in REGISTERS, one
// entry per lane, sorted (insert = ballot + popcount + one shuffle).  Per workgroup the 16 wave lists
// are merged in LDS and written out; the last workgroup to arrive (device-scope fence + counter)
// merges the per-workgroup lists the same way and writes the query's results.  Device-gated: the kernel
// reads the number of flagged queries and exits at once when it is 0 (one empty dispatch, ~5 us).
// k > 64 runs as ceil(k / 64) rounds; round r only admits rows strictly after the last result of
// round r - 1 in the (distance, row) order.
constexpr int kExactThreads = 1024;
constexpr int kExactWaves = kExactThreads / 64;
constexpr int kExactRound = 64;

struct ExactArgs {
    const float *docs;         // f32 [n][d], or null with
    const _Float16 *docs16;    // f16 [n][d]
    const float *doc_sq;
    uint32_t n_rows;
    int d;
    int metric;
    const double *q;           // [b][d]
    const double *q_sq;
    const double *q_norm;
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

// lane i of the wave holds the i-th best entry (i < cnt); every argument is wave-uniform
__device__ __forceinline__ void exact_wave_insert(double nd, uint32_t nr, int kk, int lane, double &my_d, uint32_t &my_r,
                                                  int &cnt) {
    const unsigned long long before = __ballot(lane < cnt && dist_before(my_d, my_r, nd, nr));
    const int pos = __popcll(before);  // sorted: the entries before the new one are lanes [0, pos)
    if (pos >= kk) return;
    const double up_d = __shfl_up(my_d, 1, 64);
    const uint32_t up_r = __shfl_up(my_r, 1, 64);
    if (lane > pos) {
        my_d = up_d;
        my_r = up_r;
    } else if (lane == pos) {
        my_d = nd;
        my_r = nr;
    }
    cnt = cnt < kk ? cnt + 1 : kk;
}

// The workgroup's 16 wave lists -> rank of this thread's entry among all valid ones (-1: none)
__device__ __forceinline__ int exact_block_rank(double my_d, uint32_t my_r, int cnt, double *s_d, uint32_t *s_r, int *s_cnt,
                                                int tid, int *total_out) {
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();  // previous use of the staging arrays is over
    s_d[tid] = my_d;
    s_r[tid] = my_r;
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    int total = 0, rank = 0;
    const bool valid = lane < cnt;
    for (int w = 0; w < kExactWaves; ++w) {
        const int c = s_cnt[w];
        total += c;
        if (valid)
            for (int l = 0; l < c; ++l) rank += dist_before(s_d[w * 64 + l], s_r[w * 64 + l], my_d, my_r) ? 1 : 0;
    }
    *total_out = total;
    return valid ? rank : -1;
}

// (round 3: a device function - exact_pass_kernel of vec_kernels_exact.h runs it when ONE or TWO queries were handed over,
// the batched pass otherwise)
__device__ __forceinline__ void exact_topk_serial(const ExactArgs &a, const int nf) {
    __shared__ double s_d[kExactThreads];
    __shared__ uint32_t s_r[kExactThreads];
    __shared__ int s_cnt[kExactWaves];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = min(kExactRound, a.k - kExactRound * a.round);
    const uint32_t G = gridDim.x;
    for (int f = 0; f < nf; ++f) {
        const int qi = a.flagged[f];
        const double *q = a.q + (size_t)qi * a.d;
        const double q_sq = a.q_sq[qi], q_norm = a.q_norm[qi];
        const bool bounded = a.round > 0;
        const double b_d = bounded ? a.bound_dist[qi] : 0.0;
        const uint32_t b_r = bounded ? a.bound_row[qi] : 0u;
        double my_d = 0.0;
        uint32_t my_r = 0;
        int cnt = 0;
        for (uint32_t row = blockIdx.x * kExactWaves + wave; row < a.n_rows; row += G * kExactWaves) {
            double rv;
            const double dist = a.docs16 ? exact_metric_wave(a.docs16 + (size_t)row * a.d, q, a.d, a.metric, a.doc_sq[row], q_sq, q_norm, lane, &rv)
                                         : exact_metric_wave(a.docs + (size_t)row * a.d, q, a.d, a.metric, a.doc_sq[row], q_sq, q_norm, lane, &rv);
            if (bounded && !dist_before(b_d, b_r, dist, row)) continue;
            exact_wave_insert(dist, row, kk, lane, my_d, my_r, cnt);
        }
        int total;
        int rank = exact_block_rank(my_d, my_r, cnt, s_d, s_r, s_cnt, tid, &total);
        uint64_t *mine = a.part + ((size_t)f * G + blockIdx.x) * a.list_stride * 2;
        if (rank >= 0 && rank < kk) {
            mine[2 * rank] = (uint64_t)__double_as_longlong(my_d);
            mine[2 * rank + 1] = (1ull << 32) | my_r;
        }
        if (tid < kk && tid >= total) {
            mine[2 * tid] = 0;
            mine[2 * tid + 1] = 0;
        }
        __threadfence();  // this workgroup's list is visible device-wide before its arrival is
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&a.arrive[f], 1u) == G - 1;
        __syncthreads();
        if (!s_last) continue;  // uniform per workgroup
        __threadfence();
        // ---- the last workgroup: merge the G lists (each sorted, invalid entries last) ----
        my_d = 0.0;
        my_r = 0;
        cnt = 0;
        for (uint32_t g = wave; g < G; g += kExactWaves) {
            const uint64_t *l = a.part + ((size_t)f * G + g) * a.list_stride * 2;
            for (int j = 0; j < kk; ++j) {
                const uint64_t w1 = __hip_atomic_load(l + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(w1 >> 32)) break;
                const double dj = __longlong_as_double((long long)__hip_atomic_load(l + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const int before = cnt;
                const double worst_d = __shfl(my_d, kk - 1, 64);
                const uint32_t worst_r = __shfl(my_r, kk - 1, 64);
                if (before == kk && !dist_before(dj, (uint32_t)w1, worst_d, worst_r)) break;  // the rest of this list is worse still
                exact_wave_insert(dj, (uint32_t)w1, kk, lane, my_d, my_r, cnt);
            }
        }
        rank = exact_block_rank(my_d, my_r, cnt, s_d, s_r, s_cnt, tid, &total);
        const int kout = total < kk ? total : kk;
        if (rank >= 0 && rank < kout) {
            const size_t o = (size_t)qi * a.k + (size_t)kExactRound * a.round + rank;
            if (a.out_row) a.out_row[o] = a.row_offset + (int64_t)my_r;
            if (a.out_dist) a.out_dist[o] = my_d;
            if (a.out_doc) a.out_doc[o] = a.doc_ids ? a.doc_ids[my_r] : 0;
            if (a.out_chunk) a.out_chunk[o] = a.chunk_ids ? a.chunk_ids[my_r] : (int64_t)my_r;
            if (rank == kout - 1) {
                a.bound_dist[qi] = my_d;
                a.bound_row[qi] = my_r;
            }
        }
        if (tid == 0) {
            a.arrive[f] = 0;
            if (a.round == 0) {
                if (a.out_count) a.out_count[qi] = (int)((uint32_t)a.k < a.n_rows ? (uint32_t)a.k : a.n_rows);
                if (a.out_flags) a.out_flags[qi] = MIR_FLAG_EXACT_PASS;
            }
        }
    }
}

// every query takes the exact pass (k beyond the scan's candidate lists)
// grid = b: block i hands query i over (position i of the flagged list)
__global__ __launch_bounds__(256) void flag_all_kernel(int b, int32_t *__restrict__ nflag, int32_t *__restrict__ flagged,
                                                       const double *__restrict__ q, int d, int metric, const double *__restrict__ q_norm,
                                                       double *__restrict__ qt) {
    const int i = blockIdx.x;
    if (threadIdx.x == 0) {
        flagged[i] = i;
        if (i == 0) *nflag = b;
    }
    exact_publish_query(qt, i, q + (size_t)i * d, d, metric, q_norm[i], threadIdx.x, 256);
}

// ---------------------------------------------------------------- sample thresholds
// After a scan of a SAMPLE of the rows: the klist-th best key of the sample is a lower bound on
// the klist-th best key of the whole index (the sample is a subset), so it is a valid starting
// threshold for every list of the full scan.  One block per query.
__global__ __launch_bounds__(256) void sample_threshold_kernel(const float *__restrict__ part, int nwg, int qpw,
                                                               int klist, int b_in_launch,
                                                               unsigned long long *__restrict__ gthr) {
    constexpr int kMaxVals = 2 * 256;  // two half-lanes per workgroup, kSampleWgs <= 256
    __shared__ __attribute__((aligned(16))) float vals[kMaxVals];
    __shared__ float thr;
    const int tid = threadIdx.x, q = blockIdx.x;
    if (q >= b_in_launch) return;
    const int n = 2 * nwg;
    for (int e = tid; e < kMaxVals; e += 256) {
        float v = -__builtin_inff();
        if (e < n) v = part[((size_t)(e >> 1) * qpw + q) * 2 + (e & 1)];
        vals[e] = (v == v) ? v : -__builtin_inff();
    }
    // bitonic sort, descending, one compare-exchange per thread and step (45 steps of ~100 cycles; counting, for
    // every value, how many of the 512 are greater took three times as long)
    for (int size = 2; size <= kMaxVals; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            const int lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
            const bool desc = (lo & size) == 0;
            const float x = vals[lo], y = vals[hi];
            if ((x < y) == desc) {
                vals[lo] = y;
                vals[hi] = x;
            }
        }
    __syncthreads();
    // the klist-th largest; the lowest key of that value is still a lower bound whatever the row
    if (tid == 0 && klist <= kMaxVals) {
        thr = vals[klist - 1];
        if (thr > -__builtin_inff()) gthr[q] = (unsigned long long)orderable(thr) << 32;
    }
}

// ---------------------------------------------------------------- shard merge

// One block of 64 threads per query: merge s lists of <= k (dist, row) into k.
// Shard sh's arrays start `sh * stride` bytes after the base pointers.
__global__ __launch_bounds__(64) void merge_topk_global_kernel(const char *__restrict__ dist_b,
                                                        const char *__restrict__ row_b,
                                                        const char *__restrict__ count_b, int s,
                                                        int64_t stride_d, int64_t stride_r, int64_t stride_c,
                                                        int b, int k, int descending,
                                                        double *__restrict__ out_dist,
                                                        int64_t *__restrict__ out_row,
                                                        int32_t *__restrict__ out_count) {
    const int qi = blockIdx.x;
    auto cnt = [&](int sh) { return reinterpret_cast<const int32_t *>(count_b + sh * stride_c)[qi]; };
    auto dst = [&](int sh, int p) { return reinterpret_cast<const double *>(dist_b + sh * stride_d)[(size_t)qi * k + p]; };
    auto rw = [&](int sh, int p) { return reinterpret_cast<const int64_t *>(row_b + sh * stride_r)[(size_t)qi * k + p]; };
    int total = 0;
    for (int sh = 0; sh < s; ++sh) total += cnt(sh);
    const int kout = total < k ? total : k;
    // rank by counting; s*k is small (<= 8*64)
    for (int e = threadIdx.x; e < s * k; e += 64) {
        const int sh = e / k, p = e - sh * k;
        if (p >= cnt(sh)) continue;
        const double dm = dst(sh, p);
        const int64_t rm = rw(sh, p);
        int rank = 0;
        for (int s2 = 0; s2 < s; ++s2) {
            const int c2 = cnt(s2);
            for (int p2 = 0; p2 < c2; ++p2) {
                if (s2 == sh && p2 == p) continue;
                const double d2 = dst(s2, p2);
                const int64_t r2 = rw(s2, p2);
                bool before;
                if (!descending) {
                    const bool n2 = d2 != d2, nm = dm != dm;
                    if (n2 || nm) before = (n2 == nm) ? r2 < rm : nm;
                    else before = d2 < dm || (d2 == dm && r2 < rm);
                } else {
                    before = d2 > dm || (d2 == dm && r2 > rm);
                }
                rank += before ? 1 : 0;
            }
        }
        if (rank < kout) {
            out_dist[(size_t)qi * k + rank] = dm;
            out_row[(size_t)qi * k + rank] = rm;
        }
    }
    if (threadIdx.x == 0) out_count[qi] = kout;
}

// The same merge with the s*k candidates staged in LDS first (s*k <= kMergeLds): the kernel above re-reads every
// candidate from global memory inside its rank loop - 2*s*k dependent loads per thread, 21 us at s = 8, k = 10,
// on the critical path of every step of an 8-GPU search (after the all-gather).
constexpr int kMergeLds = 1024;
// float64 -> uint64 whose unsigned order is the reference's: numbers ascending (-0 = +0), NaN last
__device__ __forceinline__ uint64_t merge_key(double d) {
    if (d != d) return ~0ull;
    const uint64_t u = (uint64_t)__double_as_longlong(d + 0.0);  // -0.0 + 0.0 = +0.0
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__global__ __launch_bounds__(256) void merge_topk_kernel(const char *__restrict__ dist_b, const char *__restrict__ row_b,
                                                         const char *__restrict__ count_b, int s, int64_t stride_d,
                                                         int64_t stride_r, int64_t stride_c, int b, int k, int descending,
                                                         double *__restrict__ out_dist, int64_t *__restrict__ out_row,
                                                         int32_t *__restrict__ out_count) {
    __shared__ double sd[kMergeLds];
    __shared__ uint64_t skey[kMergeLds];
    __shared__ int64_t sr[kMergeLds];
    __shared__ int scnt[kMergeLds];  // per shard (s <= s*k)
    const int qi = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int n = s * k;
    for (int sh = tid; sh < s; sh += nthr) scnt[sh] = reinterpret_cast<const int32_t *>(count_b + sh * stride_c)[qi];
    __syncthreads();
    // candidate (shard sh, position p) -> slot sh*k + p; four loads in flight per thread
    for (int e0 = tid; e0 < n; e0 += nthr * 4) {
        double dv[4];
        int64_t rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + nthr * u;
            dv[u] = 0.0; rv[u] = 0;
            if (e < n) {
                const int sh = e / k, p = e - sh * k;
                if (p < scnt[sh]) {
                    dv[u] = reinterpret_cast<const double *>(dist_b + sh * stride_d)[(size_t)qi * k + p];
                    rv[u] = reinterpret_cast<const int64_t *>(row_b + sh * stride_r)[(size_t)qi * k + p];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + nthr * u;
            if (e < n) {
                sd[e] = dv[u];
                sr[e] = rv[u];
                // descending scores (BM25; never NaN): larger first, ties to the LARGER row - invert both orders
                skey[e] = descending ? ~merge_key(dv[u]) : merge_key(dv[u]);
            }
        }
    }
    __syncthreads();
    int total = 0;
    for (int sh = 0; sh < s; ++sh) total += scnt[sh];
    const int kout = total < k ? total : k;
    for (int e = tid; e < n; e += nthr) {
        const int sh = e / k, p = e - sh * k;
        if (p >= scnt[sh]) continue;
        const uint64_t km = skey[e];
        const int64_t rm = sr[e];
        int rank = 0;
        for (int s2 = 0; s2 < s; ++s2) {
            const int c2 = scnt[s2];
            for (int p2 = 0; p2 < c2; ++p2) {
                const int e2 = s2 * k + p2;
                const uint64_t k2 = skey[e2];
                const int64_t r2 = sr[e2];
                const bool row_before = descending ? r2 > rm : r2 < rm;
                rank += (k2 < km || (k2 == km && row_before)) ? 1 : 0;  // e2 == e: equal key, same row -> 0
            }
        }
        if (rank < kout) {
            out_dist[(size_t)qi * k + rank] = sd[e];
            out_row[(size_t)qi * k + rank] = rm;
        }
    }
    if (tid == 0) out_count[qi] = kout;
}

}  // namespace mir
