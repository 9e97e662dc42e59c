// The sieve's 8-bit first stage (round 4): the same filter / scatter / select chain as vec_kernels_sieve.h, with the filter on
// v_mfma_i32_16x16x64_i8 over an int8 image of the shard - half the bytes of the bf16 hi image and half its matrix cycles.
// The bf16 filter at 256 queries per pass is limited by the chip's POWER (its matrix work beside its HBM stream hold the clock
// at 1.84-1.89 GHz, profiles/r04_sieve_mfma_pmc.md): fewer bytes and cheaper arithmetic per row are what is left.  What an
// 8-bit filter pays with is its margin - 4-5 x bfloat16's - i.e. candidates: everything below is arranged so that the margin
// is as small as a rigorous bound allows (a scale per tile and per query) and a candidate as cheap as possible (no LDS atomic
// on the emit path, one global atomic per (wave part, query) in the scatter).
//
// Image.  x ~ s_t X with one scale per 32-row TILE, s_t = (the tile's largest |x_i|) / 127, X = round(x / s_t) in [-127, 127]
// (one scale per index lists 2.3 x the candidates: the largest component of 10M rows is 1.5 x a tile's); 32-row tiles of
// KS64 * 2 blocks of 1 KiB: block (s, rh) = k-step of 64 columns s, row half rh: 64 lanes x 16 int8, lane l = (row 16 rh + (l & 15),
// columns 64 s + 16 (l >> 4) .. + 15) - the A operand of the MFMA (any in-lane order would do: the B operand is packed the
// same way and a dot product does not care).  An LDS stage is TWO tiles (64 rows, 24 KiB at d = 384).
// Queries: q ~ s_q Q with a scale per QUERY (its largest |q_i| / 127), B-operand fragments per 16 queries.
//
// Exactness.  I = X.Q is exact (int32; |I| <= 127^2 d < 2^24 for d <= 1024, so its float32 conversion is exact too).  With
// x^ = s_t X, q^ = s_q Q:
//   x.q - s_t s_q I = (x - x^).q + x^.(q - q^),  |.| <= |x - x^| |q| + (|x| + |x - x^|) |q - q^|   (Cauchy-Schwarz)
// - hihi_margin()'s formula with the int8 residuals in place of the bf16 ones: e_t = the largest |x - x^| over a TILE's rows is
// measured at build time (i8 tile parameters), |q - q^| per query at prep time; a row's margin is its tile's,
// mg_t = e_t (|q| + |q - q^|) + |x|max |q - q^| (+ slop) + 2 units (i8_margin_tile).  Squared L2
// ranks v = 2 x.q - |x|^2 (|x|^2 the float32 value the reference formula uses).  The common path compares INTEGERS: v >= bound
// implies 2 s_t s_q I >= bound_t + |x|^2 >= bound_t + amin, amin = the index's smallest squared norm, i.e. I >= ib with ib the
// integer part of (bound_t + amin) / (2 s_t s_q) rounded down - four vector instructions per tile and query tile - then one
// maximum of eight accumulators and one compare per lane, no arithmetic on the values.  Whatever passes is redone in float
// exactly as the bf16 filter does it (v = 2 s_t s_q I - |x|^2 per row, one rounding), listed with v and its margin, and goes through the scatter and select of vec_kernels_sieve.h: every listed
// row's true value lies in [v - mg, v + mg], and the rows select cannot exclude get the reference's float64 formula from the
// float32 rows.  The integer test is conservative by (|x|^2 - amin) / (2 s_t s_q) units, which is why the image is built only
// for shards whose squared norms agree to 1e-3 (normalised embeddings: to 1e-7).
//
// Served: float32 shards the bf16 sieve serves (d padded to 128 / 256 / 384, >= 32K rows) whose rows are finite and of one
// norm, all four metrics (cosine: see the kernel), k <= 16 (kI8MaxK: the lists grow with k).  Everything else - larger k,
// other norms, the wide and the float16-native shards - stays on the bf16 / float16 filters.
//
// Built by default for every shard that qualifies (MIR_SIEVE_I8=0 at index build keeps the bf16 filter: the A/B switch;
// `mir_index_scan_stats` word 6 says which an index has); the bf16 hi image is built beside it and serves k > 16.  Measured on
// 10M x 384 unit rows, 256 queries per step (profiles/r04_i8_sieve.md): 1.13-1.14 ms per step against the bf16 filter's 1.83-1.87
// (224-227k against 135-141k QPS), 1.8k + 2.4k candidates per query against 330 + 289.
#pragma once
#include "vec_kernels_sieve.h"

#ifndef I8_ABL
#define I8_ABL 0  // measurement builds only (results wrong by design): bit 0 no filter, bit 1 no barrier, bit 2 no DMA after the prologue
#endif

namespace mir {

typedef int __attribute__((ext_vector_type(4))) i32x4;

// i8 statistics (floats): [0] largest row norm, [1] largest squared norm (float32 doc_sq), [2] largest |x - x^| (information),
// [7] smallest squared norm (words 1, 2, 7 are reduced as float bits: non-negative floats order as their bits); written by the
// host once the build is complete: [4] nmin, [5] nmax = the rows' smallest / largest norm rounded down / up, [6] 1 / nmin
// rounded up (cosine)
constexpr int kI8StatWords = 8;
// tile parameters (float4 per 32-row tile): x = s_t, y = e_t (largest |x - x^| of its rows, rounded up), z = 1 / (2 s_t), w = -
constexpr int kI8MaxK = 16;           // results per query the int8 first stage serves (vec_index.hip, enqueue_search)
constexpr int kI8Region = 32768;       // candidates per launch and CU: eight wave-private parts of 4096 (one eight-wave workgroup, or two of four waves)
constexpr int kI8WavePart = kI8Region / 8;
// The margin of a row of a tile with residual bound e_t and scale s_t, in inner-product units, as mg_t = e_t * A + B + 2 s_t s_q:
//   A = (|q| + |q - q^|) (1 + 1e-5), B = (|x|max |q - q^| + 3e-5 |x|max |q|) (1 + 1e-5)  [hihi_margin's terms, regrouped by e_t]
// (x 2 in squared-L2 units).  The filter and the select kernel both compute it from these two per-query numbers.
__device__ __forceinline__ void i8_margin_ab(float qn, float eq, const float *__restrict__ stats, float &A, float &B) {
    i8_margin_ab_decl(qn, eq, stats, A, B);  // (defined in vec_kernels_sieve.h: the select kernel computes the same margin)
}
__device__ __forceinline__ float i8_margin_tile(bool l2, float A, float B, float e_t, float s_t, float sq) {
    const float m = fmaf(e_t, A, B) + 2.0f * s_t * sq;
    return l2 ? 2.0f * m : m;
}

// one block per 32-row tile: s_t from the tile's largest |x_i| (an all-zero tile: 1)
__global__ __launch_bounds__(256) void i8_tile_scale_kernel(const float *__restrict__ src, int64_t n, int d, float4 *__restrict__ tparam) {
    __shared__ float red[4];
    const int64_t tile = blockIdx.x;
    const int64_t r0 = tile * kTileRows;
    const int64_t rows = n - r0 < kTileRows ? n - r0 : kTileRows;
    const float *p = src + r0 * (int64_t)d;
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < rows * d; i += 256) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float st = m > 0.f ? m / 127.0f : 1.0f;
        tparam[tile] = make_float4(st, 0.f, 0.5f / st, 0.f);
    }
}
// statistics: [0] the index's largest norm, [7] a minimum over float bits
__global__ void i8_scale_kernel(float *__restrict__ stats, const float *__restrict__ max_norm) {
    stats[0] = max_norm[0];
    stats[7] = __builtin_inff();
}

__device__ __forceinline__ int i8_quant(float x, float inv) {
    const float r = rintf(x * inv);
    return (int)fminf(fmaxf(r, -127.0f), 127.0f);
}

// f32 [n][d] -> the int8 image.  One thread per (tile, block, lane); columns past d and rows past n are 0.
__global__ __launch_bounds__(256) void pack_i8_kernel(const float *__restrict__ src, int64_t n, int d, int ks64, int64_t total_lanes,
                                                      const float4 *__restrict__ tparam, uint4 *__restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_lanes) return;
    const int lane = (int)(gid & 63);
    const int64_t blk = gid >> 6;
    const int nb = ks64 * 2;
    const int b = (int)(blk % nb);
    const int64_t tile = blk / nb;
    const int s = b >> 1, rh = b & 1;
    const int64_t row = tile * kTileRows + 16 * rh + (lane & 15);
    const int col0 = 64 * s + 16 * (lane >> 4);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (row < n) {
        const float inv = 1.0f / tparam[tile].x;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float x = col0 + j < d ? src[row * (int64_t)d + col0 + j] : 0.f;
            w[j >> 2] |= ((uint32_t)i8_quant(x, inv) & 0xffu) << (8 * (j & 3));
        }
    }
    dst[blk * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// per row: |x - s_t X| -> its tile's maximum (tparam.y, as float bits: non-negative floats order as their bits); the index's
// largest and smallest squared norm (statistics words 1, 7).  16 lanes per row.
__global__ __launch_bounds__(256) void i8_residual_kernel(const float *__restrict__ src, int64_t n, int d, const float *__restrict__ doc_sq,
                                                          float4 *__restrict__ tparam, float *__restrict__ stats) {
    const int lg = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    double e2 = 0.0;
    if (row < n) {
        const float s = tparam[row / kTileRows].x;
        const float inv = 1.0f / s;  // (as pack_i8_kernel)
        for (int j = lg; j < d; j += 16) {
            const float x = src[row * (int64_t)d + j];
            const double r = (double)x - (double)s * (double)i8_quant(x, inv);
            e2 += r * r;
        }
    }
    e2 = group_sum<16>(e2);
    if (lg == 0 && row < n) {
        const float e = (float)sqrt(e2) * (1.0f + 1e-6f);
        atomicMax(reinterpret_cast<unsigned int *>(&tparam[row / kTileRows].y), __float_as_uint(e));
        // the index-wide extremes: an atomic only where the value read first does not already cover this row (10M rows on three
        // words took 85 ms; a stale read costs one atomic more, never a wrong extreme)
        const float a = doc_sq[row];
        const volatile float *vs = stats;
        if (e > vs[2]) atomicMax(reinterpret_cast<unsigned int *>(stats) + 2, __float_as_uint(e));
        if (a > vs[1]) atomicMax(reinterpret_cast<unsigned int *>(stats) + 1, __float_as_uint(a));
        if (a < vs[7]) atomicMin(reinterpret_cast<unsigned int *>(stats) + 7, __float_as_uint(a));
    }
}

// The filter's per-stage record (72 floats per 64-row stage): the stage's 64 squared norms (squared L2, inner product: unread) or
// inverse norms (cosine), then its two tiles' parameters - ONE contiguous piece of the stage's DMA, one address.
__global__ __launch_bounds__(128) void i8_stage_record_kernel(const float *__restrict__ col, int64_t n_col, const float4 *__restrict__ tparam,
                                                              float *__restrict__ rec) {
    const int64_t stage = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= 72) return;
    float v;
    if (t < 64) {
        const int64_t row = stage * 64 + t;
        v = row < n_col ? col[row] : 0.f;
    } else {
        v = reinterpret_cast<const float *>(tparam)[stage * 8 + (t - 64)];
    }
    rec[stage * 72 + t] = v;
}

// ---- queries ----
// Kernel 1, one block per query (+ the blocks that zero the control words, as prep_queries16_kernel): q_sq and q_norm in float64
// (the same sums in the same order as prep_queries16_kernel: the select kernel's distances use them) and the query's largest |q_i|.
__global__ __launch_bounds__(64) void prep_queries_i8_stats_kernel(const double *__restrict__ q, int b, int d, double *__restrict__ q_sq,
                                                                   double *__restrict__ q_norm, float *__restrict__ q_amax,
                                                                   unsigned long long *__restrict__ gthr, int gthr_words) {
    const int lane = threadIdx.x, blk = blockIdx.x;
    if (gthr && blk * 64 + lane < gthr_words) gthr[blk * 64 + lane] = 0;
    const int qi = blk;
    if (qi >= b) return;
    double s = 0.0;
    float m = 0.f;
    bool bad = false;
    for (int j = lane; j < d; j += 64) {
        const double x = q[(int64_t)qi * d + j];
        s += x * x;
        const float a = fabsf((float)x);
        bad |= !(a < __builtin_inff());  // NaN or infinite
        m = fmaxf(m, a);
    }
    s = wave_sum(s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    bad = __any(bad);
    if (lane == 0) {
        q_sq[qi] = s;
        q_norm[qi] = sqrt(s);
        q_amax[qi] = bad ? -1.0f : m;  // (a query with a NaN / an infinity takes no part in the scale; the filter passes everything for it)
    }
}
// Kernel 2.  Blocks [0, ntiles16 * ks64): B-operand fragments of query tile w, k-step s: lane l = (query 16 w + (l & 15), columns
// 64 s + 16 (l >> 4) .. + 15), every query with its own scale.  Blocks [ntiles16 * ks64, + b): the query's scale and its
// residual |q - s_q Q| in float64.
__global__ __launch_bounds__(64) void prep_queries_i8_quant_kernel(const double *__restrict__ q, int b, int d, int ks64, int ntiles16,
                                                                   const float *__restrict__ q_amax, uint4 *__restrict__ qfrag,
                                                                   double *__restrict__ q_err, float *__restrict__ q_scale) {
    const int lane = threadIdx.x, blk = blockIdx.x;
    if (blk < ntiles16 * ks64) {
        const int s = blk % ks64, w = blk / ks64;
        const int qi = 16 * w + (lane & 15);
        const int col0 = 64 * s + 16 * (lane >> 4);
        uint32_t wv[4] = {0u, 0u, 0u, 0u};
        const float m = qi < b ? q_amax[qi] : -1.0f;
        if (m >= 0.f) {
            const float inv = m > 0.f ? 127.0f / m : 1.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float x = col0 + j < d ? (float)q[(int64_t)qi * d + col0 + j] : 0.f;
                wv[j >> 2] |= ((uint32_t)i8_quant(x, inv) & 0xffu) << (8 * (j & 3));
            }
        }
        qfrag[(int64_t)blk * 64 + lane] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    } else {
        const int qi = blk - ntiles16 * ks64;
        if (qi >= b) return;
        const float m = q_amax[qi];
        const bool ok = m >= 0.f;
        const float inv = m > 0.f ? 127.0f / m : 1.0f;   // (the same value the fragment blocks use)
        const float sq = 1.0f / inv;                     // the scale the filter and select multiply with: Q was made with `inv`,
        double e2 = 0.0;                                 // the residual below is measured against THIS sq
        for (int j = lane; j < d; j += 64) {
            const double x = q[(int64_t)qi * d + j];
            const double r = ok ? x - (double)sq * (double)i8_quant((float)x, inv) : x;  // ((float)x: what the fragment was made from)
            e2 += r * r;
        }
        e2 = wave_sum(e2);
        if (lane == 0) {
            q_err[qi] = sqrt(e2);
            q_scale[qi] = sq;
        }
    }
}

// ---- the filter ----
// Geometry: NW waves per workgroup.  8 waves: one workgroup per CU, six 24-KiB stages.  4 waves (each with twice the query tiles):
// TWO workgroups per CU with three stages each - one workgroup's barrier and tile boundary are the other's matrix time (the
// barrier alone was 12 % of the 8-wave launch), and every document fragment read from LDS feeds twice the MFMAs.
__host__ __device__ constexpr int sieve_i8_stages(int nw) { return nw == 8 ? kSieveStages : 3; }
__host__ __device__ constexpr size_t sieve_i8_lds_bytes(int ks64, int nw) { return (size_t)sieve_i8_stages(nw) * (ks64 * 4 * 1024 + (64 + 8) * 4) + 64; }

// QT = query tiles (16 queries each) per wave, as sieve_q16_kernel.  KS64 = k-steps of 64 columns (2, 4, 6).
// Candidates: every wave writes its own part of the workgroup's region (kI8WavePart entries; the count is wave-uniform: no
// LDS atomic on the emit path), ccount[workgroup][8].
template <int KS64, int KIND, bool SAMPLE, int QT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void sieve_i8_kernel(const uint4 *__restrict__ docs, const float *__restrict__ aux,
                                                          const uint4 *__restrict__ qfrag, const double *__restrict__ q_norm,
                                                          const double *__restrict__ q_sq, const double *__restrict__ q_err,
                                                          const float *__restrict__ stats, const float *__restrict__ q_scale,
                                                          uint32_t n_rows, uint32_t stage0, uint32_t n_stages, int nq, int nan_guard,
                                                          const uint64_t *__restrict__ gthr, uint64_t *__restrict__ cand,
                                                          float *__restrict__ candv, uint32_t *__restrict__ ccount,
                                                          float *__restrict__ part_sample, unsigned long long *__restrict__ stat) {
    static_assert(KIND == SCAN_L2 || KIND == SCAN_IP || KIND == SCAN_COS, "scan kind");
    static_assert(QT == 1 || QT == 2 || QT == 4, "query tiles per wave");
    static_assert(NW == 8 || NW == 4, "waves per workgroup");
    constexpr int NS = sieve_i8_stages(NW);
    constexpr int APW = 64 / NW;           // squared norms a wave brings per stage
    constexpr int NB = KS64 * 2;           // 1-KiB blocks per 32-row tile
    constexpr int SB = NB * 2;             // per stage (two tiles)
    constexpr int STAGE_U4 = SB * 64;
    constexpr int PPW = SB / NW;           // 1-KiB DMA pieces per wave per stage
    // the stage's record travels with it: 64 squared norms (inverse norms for cosine; inner product never reads them) and its two
    // tiles' parameters, one DMA instruction per wave
    constexpr int PW = PPW + 1;
    constexpr int AS = 64 + 8;
    constexpr int D = NS - 1;
    constexpr int QPL = NW * 16 * QT;
    static_assert(SB % NW == 0, "sieve_i8: d padded to a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem);
    float *aux_lds = reinterpret_cast<float *>(smem + (size_t)NS * STAGE_U4 * 16);

    const int tid = threadIdx.x, lane = tid & 63, qc = lane & 15, jg = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x;

    const float amin = KIND == SCAN_L2 ? stats[7] : 0.f;  // the smallest squared norm of the index
    int qloc[QT];
    bool lane_live[QT];
    unsigned long long live_mask[QT];
    constexpr float L = KIND == SCAN_L2 ? 2.0f : 1.0f;  // ranking units per inner-product unit
    // Cosine ranks x.q / |x| on a shard whose norms all lie in [nmin, nmax] (statistics words 4, 5; the image is built only where
    // they agree to 5e-4): true cosine >= T implies x.q >= T |x| >= min(T nmin, T nmax), so the INTEGER test is the inner
    // product's with that threshold; the float re-test and the listed value are in cosine units (v = s_t s_q I / |x|, the
    // row's inverse norm from the aux column), with the inner-product margin times the largest inverse norm (word 6).
    const float ax_hi = KIND == SCAN_COS ? stats[6] : 1.0f;
    float tbc[QT];  // (cosine) the threshold in its own units
    // per query: its scale, the margin's two coefficients (i8_margin_ab), the threshold with its rounding slack, and the two
    // numbers the per-tile integer bound is made of: ib_t = floor((P1 - e_t A1) / s_t) - 4 (see the header)
    // (the float bound of the rare path is rebuilt from the same three numbers: bound_t = (P1 - e_t A1 - 2 s_t) L sq - amin)
    float sq[QT], mA[QT], mB[QT], P1[QT], A1[QT], guard[QT], best[QT];
    i32x4 qh[QT][KS64];
    const bool active = nq > wave8 * QT * 16;
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int t16 = wave8 * QT + u;
        qloc[u] = t16 * 16 + qc;
        lane_live[u] = qloc[u] < nq;
        live_mask[u] = __builtin_amdgcn_ballot_w64(lane_live[u]);
        sq[u] = 1.0f; mA[u] = 0.f; mB[u] = 0.f; P1[u] = -__builtin_inff(); A1[u] = 0.f; tbc[u] = -__builtin_inff();
        guard[u] = __builtin_inff(); best[u] = -__builtin_inff();
        if (lane_live[u]) {
            const float qn = (float)q_norm[qloc[u]] * (1.0f + 1e-6f);
            const float eq = (float)q_err[qloc[u]] * (1.0f + 1e-6f);
            const bool good = qn < __builtin_inff() && eq < __builtin_inff();  // (false for a NaN too: such a query passes everything, as in the bf16 filter)
            sq[u] = q_scale[qloc[u]];
            if (good) i8_margin_ab(qn, eq, stats, mA[u], mB[u]);
            if (!SAMPLE) {
                const uint64_t key = gthr[qloc[u]];
                if (key != 0 && good) {
                    const float t = key_value(key);
                    float tb = t - 2e-6f * fabsf(t);
                    tbc[u] = tb;
                    if (KIND == SCAN_COS) tb = fminf(tb * stats[4], tb * stats[5]) - 1e-6f * fabsf(tb) * stats[5];
                    P1[u] = (tb + amin - L * mB[u]) / (L * sq[u]);
                    A1[u] = mA[u] / sq[u];
                    if (!(P1[u] == P1[u]) || !(A1[u] == A1[u])) { P1[u] = -__builtin_inff(); A1[u] = 0.f; tbc[u] = -__builtin_inff(); }
                }
            } else if (nan_guard) {
                const float qs = (float)q_sq[qloc[u]];
                guard[u] = qs - 1e-5f * fabsf(qs);
            }
        }
        const uint4 *qs = qfrag + (size_t)t16 * KS64 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS64; ++s) qh[u][s] = __builtin_bit_cast(i32x4, qs[s * 64]);
    }
    const uint32_t my_stages = blockIdx.x < n_stages ? (n_stages - blockIdx.x + G - 1) / G : 0;
    const uint32_t NG = my_stages;

    auto issue = [&](uint32_t g) {
        const uint32_t stage = stage0 + blockIdx.x + g * G;
        const uint4 *src = docs + (size_t)stage * STAGE_U4 + (wave8 * PPW) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(ring) + ((g % NS) * STAGE_U4 + (wave8 * PPW) * 64) * 16);
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_b128(src + i * 64, dst + i * 1024);
        {   // the stage's record (i8_stage_record_kernel): every wave its share of the 64 norms, the last one the 8 parameter floats behind them
            // (a wave-uniform base and the lane's byte offset: a 64-bit per-lane address would be one more live register pair in a
            // kernel that has none to spare - spilled, its reload's `s_waitcnt vmcnt(0)` drained the DMA ring at every stage)
            const uint32_t adst = __builtin_amdgcn_readfirstlane(lds_addr_of(aux_lds) + ((g % NS) * AS + wave8 * APW) * 4);
            const float *abase = aux + (size_t)stage * AS + wave8 * APW;
            if (lane < APW + (wave8 == NW - 1 ? 8 : 0)) glds4_b32_sv(abase, (uint32_t)lane * 4u, adst);
        }
    };
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int s = 0; s < KS64; ++s) asm volatile("" : "+v"(qh[u][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t g = 0; g < (uint32_t)D && g < NG; ++g) issue(g);

    auto wait_stage = [&](uint32_t g) {
        const uint32_t younger = (NG - 1 - g) < (uint32_t)(D - 1) ? (NG - 1 - g) : (uint32_t)(D - 1);
        if (younger == (uint32_t)(D - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    uint64_t *region = cand + ((size_t)blockIdx.x * NW + wave8) * kI8WavePart;
    float *regionv = candv + ((size_t)blockIdx.x * NW + wave8) * kI8WavePart;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t wcount = 0;  // (wave-uniform) candidates this wave has written; beyond kI8WavePart they are counted, not stored

    // The filter of one 32-row tile and query tile, one tile late (as sieve_q16_kernel's): the tile's integer bound from its scale
    // and residual (four vector instructions), eight int32 accumulators against it - a maximum and ONE compare; what passes is
    // redone in float.  tp = the tile's parameters (s_t, e_t, 1 / (2 s_t))
    auto filter = [&](int u, const i32x4 &c0, const i32x4 &c1, const float (&ax)[8], uint32_t t, const float4 &tp) {
        if (SAMPLE) {
            if (lane_live[u]) {
                const float vs = L * tp.x * sq[u], mg = i8_margin_tile(KIND == SCAN_L2, mA[u], mB[u], tp.y, tp.x, sq[u]) * ax_hi;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float fi = (float)(r < 4 ? c0[r & 3] : c1[r & 3]);
                    const float w = KIND == SCAN_L2 ? fmaf(vs, fi, -ax[r]) : KIND == SCAN_COS ? vs * fi * ax[r] : vs * fi;
                    if (w + mg < guard[u]) best[u] = fmaxf(best[u], w - mg);
                }
            }
            return;
        }
        const float x2 = fmaf(fmaf(-tp.y, A1[u], P1[u]), 2.0f * tp.z, -4.0f);
        const int ib = (int)__builtin_amdgcn_fmed3f(floorf(x2), -2.0e9f, 2.0e9f);  // (no threshold / a bad query: -inf -> everything passes)
        const int m0 = max(max(c0[0], c0[1]), c0[2]), m1 = max(max(c0[3], c1[0]), c1[1]);
        const int m = max(max(max(c1[2], c1[3]), m0), m1);
        if ((__builtin_amdgcn_ballot_w64(m >= ib) & live_mask[u]) == 0ull) return;
        asm volatile("" : "+s"(t));  // (the rare path below depends on t: nothing of it is computed ahead of the branch)
        const float vs = L * tp.x * sq[u];
        // tb - mg_t in the ranking's units, from the numbers the integer bound was made of (one more rounding or two than
        // i8_margin_tile's own arithmetic: the factor below widens it by more than that; P1 = -inf: everything passes)
        float bound = (fmaf(-tp.y, A1[u], P1[u]) - 2.0f * tp.x) * (L * sq[u]) - amin;
        if (KIND == SCAN_COS) bound = tbc[u] - i8_margin_tile(false, mA[u], mB[u], tp.y, tp.x, sq[u]) * ax_hi;  // (in cosine units; tbc = -inf: everything passes)
        const float bound_w = bound - 4e-6f * (fabsf(bound) + amin);
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float fi = (float)(r < 4 ? c0[r & 3] : c1[r & 3]);
            v[r] = KIND == SCAN_L2 ? fmaf(vs, fi, -ax[r]) : KIND == SCAN_COS ? vs * fi * ax[r] : vs * fi;
        }
        uint32_t pm = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) pm |= (uint32_t)(!(v[r] < bound_w)) << r;
        if (!lane_live[u]) pm = 0;
        if (!__any(pm != 0)) return;
        const uint32_t row0 = t * kTileRows + 4 * jg;  // this lane's rows: row0 + 16 rh + i
        if (t * kTileRows + kTileRows > n_rows) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (row0 + 16 * (r >> 2) + (r & 3) >= n_rows) pm &= ~(1u << r);
        }
        while (__any(pm != 0)) {
            const bool has = pm != 0;
            const int r = has ? __builtin_ctz(pm) : 0;
            const unsigned long long bal = __ballot(has);
            const uint32_t slot = wcount + (uint32_t)__popcll(bal & lt_mask);
            wcount += (uint32_t)__popcll(bal);
            float vr = v[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) vr = r == i ? v[i] : vr;
            if (has && slot < (uint32_t)kI8WavePart) {
                region[slot] = ((uint64_t)(uint32_t)qloc[u] << 32) | (uint64_t)(row0 + 16 * (r >> 2) + (r & 3));
                regionv[slot] = vr;
            }
            pm &= pm - 1;
        }
    };

    i32x4 p0[QT], p1[QT];  // the previous tile's accumulators, squared norms and index
#pragma unroll
    for (int u = 0; u < QT; ++u) { p0[u] = i32x4{0, 0, 0, 0}; p1[u] = i32x4{0, 0, 0, 0}; }
    float pax[8] = {};
    float4 ptp = make_float4(1.f, 0.f, 0.5f, 0.f);
    uint32_t pt = 0;
    bool have_prev = false;
    for (uint32_t g = 0; g < my_stages; ++g) {
        const uint32_t stage = stage0 + blockIdx.x + g * G;
#if !(I8_ABL & 4)
        wait_stage(g);
#endif
#if !(I8_ABL & 2)
        __builtin_amdgcn_s_barrier();
#endif
        if (!active) {
            if (g + D < NG) issue(g + D);
            continue;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint4 *st = ring + (size_t)(g % NS) * STAGE_U4 + sub * (NB * 64) + lane;
            constexpr int PF = QT == 4 ? 1 : (2 < KS64 ? 2 : KS64 - 1);  // fragment pairs requested ahead of their MFMAs (four query tiles: eight MFMAs per pair already)
            uint4 f0[PF + 1], f1[PF + 1];
#pragma unroll
            for (int i = 0; i < PF; ++i) { f0[i] = st[(2 * i + 0) * 64]; f1[i] = st[(2 * i + 1) * 64]; }
            const float4 ctp = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * AS + 64 + 4 * sub);  // this tile's parameters
            float cax[8] = {};
            if (KIND != SCAN_IP) {  // rows 32 sub + 16 rh + 4 jg + i of this stage: squared norms (squared L2) / inverse norms (cosine)
                const float4 a0 = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * AS + 32 * sub + 4 * jg);
                const float4 a1 = *reinterpret_cast<const float4 *>(aux_lds + (g % NS) * AS + 32 * sub + 16 + 4 * jg);
                cax[0] = a0.x; cax[1] = a0.y; cax[2] = a0.z; cax[3] = a0.w;
                cax[4] = a1.x; cax[5] = a1.y; cax[6] = a1.z; cax[7] = a1.w;
            }
            i32x4 c0[QT], c1[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) { c0[u] = i32x4{0, 0, 0, 0}; c1[u] = i32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int s = 0; s < KS64; ++s) {
                if (s + PF < KS64) {
                    f0[(s + PF) % (PF + 1)] = st[(2 * (s + PF) + 0) * 64];
                    f1[(s + PF) % (PF + 1)] = st[(2 * (s + PF) + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    c0[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, f0[s % (PF + 1)]), qh[u][s], c0[u], 0, 0, 0);
                    c1[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, f1[s % (PF + 1)]), qh[u][s], c1[u], 0, 0, 0);
                }
                if (sub == 0 && s == 0) {  // the next stage's DMA once the matrix pipe has work queued
                    __builtin_amdgcn_sched_barrier(0);
#if !(I8_ABL & 4)
                    if (g + D < NG) issue(g + D);
#endif
                }
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    if (s == 1 + u && s < KS64 && have_prev) {  // the previous tile's filter, one query tile per k-step
                        __builtin_amdgcn_sched_barrier(0);
#if I8_ABL & 1
                        if (g == 0xffffff)
#endif
                        filter(u, p0[u], p1[u], pax, pt, ptp);
                    }
            }
            if (KS64 < QT + 1 && have_prev) {  // (few k-steps per tile: the last query tiles' filters did not fit above)
#pragma unroll
                for (int u = KS64 - 1; u < QT; ++u) filter(u, p0[u], p1[u], pax, pt, ptp);
            }
#pragma unroll
            for (int u = 0; u < QT; ++u) { p0[u] = c0[u]; p1[u] = c1[u]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) pax[i] = cax[i];
            ptp = ctp;
            pt = stage * 2 + sub;
            have_prev = true;
        }
    }
    if (have_prev) {
#pragma unroll
        for (int u = 0; u < QT; ++u) filter(u, p0[u], p1[u], pax, pt, ptp);
    }
    if (SAMPLE) {
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float o = __shfl_xor(best[u], 16, 64);
            const float b2 = fmaxf(best[u], o);
            if (lane_live[u] && (jg == 0 || jg == 2)) part_sample[((size_t)blockIdx.x * QPL + qloc[u]) * 2 + (jg >> 1)] = b2;
        }
        return;
    }
    if (lane == 0) {
        ccount[(size_t)blockIdx.x * NW + wave8] = wcount;  // may exceed kI8WavePart: the scatter then hands every query to the exact pass
        if (stat && wcount) atomicAdd(stat, (unsigned long long)wcount);
    }
}

// The scatter behind the int8 filter: a workgroup's candidates lie in eight wave-private parts.  One block per (workgroup, part):
// count the part's candidates per query in LDS, reserve each query's range of its list with ONE atomic, then place them.
// (A thread and an atomic per candidate, sieve_scatter_kernel's way, took 157 us per launch at 8.8k candidates per query.)
__global__ __launch_bounds__(256) void sieve_scatter_i8_kernel(SieveScatterArgs a) {
    __shared__ uint32_t s_cnt[256], s_base[256];
    const int tid = threadIdx.x;
    const int part = blockIdx.x;  // (region x 8 + wave)
    uint32_t cnt = a.ccount[part];
    s_cnt[tid] = 0;
    __syncthreads();
    if (cnt > (uint32_t)kI8WavePart) {  // the part overflowed: whose candidates were lost is unknown
        for (int i = tid; i < a.nq; i += 256) a.l.over[a.q0 + i] = 1;
        cnt = kI8WavePart;
    }
    const uint64_t *region = a.cand + (size_t)part * kI8WavePart;
    const float *regionv = a.candv + (size_t)part * kI8WavePart;
    for (uint32_t e = tid; e < cnt; e += 256) atomicAdd(&s_cnt[(int)(region[e] >> 32) & 255], 1u);
    __syncthreads();
    {
        const uint32_t c = s_cnt[tid];
        s_base[tid] = c ? atomicAdd(&a.l.count[(size_t)(a.q0 + tid) * kSieveCountStride], c) : 0u;
        s_cnt[tid] = 0;
    }
    __syncthreads();
    for (uint32_t e = tid; e < cnt; e += 256) {
        const uint64_t key = region[e];
        const int ql = (int)(key >> 32) & 255, qi = a.q0 + ql;
        const uint32_t slot = s_base[ql] + atomicAdd(&s_cnt[ql], 1u);
        if (slot < (uint32_t)kSieveQueryCap) {
            a.l.row[(size_t)qi * kSieveQueryCap + slot] = (uint32_t)key;
            a.l.rv[(size_t)qi * kSieveQueryCap + slot] = regionv[e];
        } else {
            a.l.over[qi] = 1;
        }
    }
}

}  // namespace mir
