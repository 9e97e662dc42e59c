// The 128-query scan of the float32 index, round 2: 16 queries per wave over the FULL dimension, 16x16x32 MFMA tiles,
// the bf16 correction products only where a block can reach a threshold.
//
// Why (profiles/r01_scan_clock_pmc.md): round 1's K-split kernel (scan_topk_b128_kernel, two waves per
// 32-query tile, each half of K) streamed 10M x 384 rows in an almost
// constant number of shader cycles whatever the batch, but its clock falls from 1.94 GHz (32 queries) to 1.53 GHz
// (128): it is power-limited, and the power goes into three bf16 MFMAs per fragment pair.  Two of the three are
// corrections (hi*lo, lo*hi) that matter only for rows near a query's threshold: the hi*hi product alone bounds a
// row's value to within 8e-3*|d||q| (kHiHiRelErr), enough to decide that NO row of a 32-row block can reach ANY of a wave's 16
// thresholds.  The K-split kernel could not use that: a wave pair knows its combined hi*hi score one tile late, when
// the tile's fragments have left the LDS ring.  Here a wave owns 16 queries over all of K (v_mfma_f32_16x16x32_bf16:
// the query tile is 16 columns, so the fragments of 16 queries x 384 dimensions are the same 96 VGPRs a K-half of 32
// queries was), so it has the complete hi*hi score in the stage that streams the tile's hi blocks and runs the
// corrections - in the next stage, which streams the tile's lo blocks - only if some lane's score + margin reaches
// its threshold (a wave-uniform vote).  Every value that enters a candidate list is still the full bf16x3 sum, so the
// lists, the completeness check of finalize_kernel and its bound (scan_rel_err) are unchanged.  The 16x16x32 shape also holds a
// higher clock than 32x32x16 under load (MI355X_MICROARCH.md, DVFS give-back item 7).
//
// Index image ("layout16", built by pack_split16_f32_kernel for d padded to 128 / 256 / 384): per 32-row tile, first
// all hi blocks then all lo blocks; block (s, rh) = k-step of 32 columns s, row half rh: 64 lanes x 8 bf16, lane l =
// (row 16 rh + (l & 15), columns 32 s + 8 (l >> 4) .. + 7): the A operand of v_mfma_f32_16x16x32_bf16.  A stage of the
// LDS-DMA ring is half a tile: stage 2t = tile t's hi blocks, stage 2t + 1 its lo blocks.
//
// Candidate buffers: one per QUERY (a query belongs to one wave) in LDS, klist kept entries plus room for 16 (at least
// 4) appended ones; the four lanes that hold a query's column (lane & 15) keep its threshold and fill count replicated.
// A passing value is APPENDED (all lanes at once: a column's up to four appends per round get consecutive slots from a
// wave vote); when a column's buffer would overflow, the wave compacts it - every lane takes one entry, counts the
// entries above it and the best klist go back in order, the klist-th becoming the column's threshold.  (Handling
// candidates one at a time with a wave-wide minimum search per replacement was fine for seeded 10M-row shards and cost
// 0.3 ms on a 5k-row index at 16 queries, where every row is a candidate.)
#pragma once
#include "vec_kernels.h"

namespace mir {

typedef float __attribute__((ext_vector_type(4))) f32x4;

// |x.q - bf16(x).bf16(q)| <= (u + u + u^2) sum|x_i q_i| <= 7.83e-3 |x||q| with u = 2^-8, bfloat16's unit roundoff (8 significant
// bits, round to nearest even); the float32 accumulation of K <= 384 exact products adds < 2.5e-5.  (Until late in round 3 this
// was 4e-3 - u taken as 2^-9: true of random data many times over, not of the worst case; tests/test_gpu_sieve.py
// ::test_worst_case_bf16_rounding builds the input that needs the whole bound.)
#ifndef MIR_HIHI_REL_ERR
#define MIR_HIHI_REL_ERR 8.0e-3f  // (the q16 list scan's constant margin; the sieve's is hihi_margin() below)
#endif
constexpr float kHiHiRelErr = MIR_HIHI_REL_ERR;
#ifndef MIR_MARGIN_SCALE
#define MIR_MARGIN_SCALE 1.0f  // (a build with 0.5f must FAIL test_worst_case_bf16_rounding: tools/worst_case_margin_check.sh)
#endif

// The sieve's margin, from what THIS index and THIS query actually lose to bfloat16 instead of the worst case: with
// x = hx + dx, q = hq + dq:  x.q - hx.hq = dx.q + hx.dq, so |x.q - hx.hq| <= |dx||q| + (|x| + |dx|)|dq| (Cauchy-Schwarz) - rigorous
// for every row once |dx| is replaced by its maximum over the rows (norm statistics word 2, bf16_residual_norm) and |x| by
// the largest norm (word 0); cosine ranks x.q / |x|: |dx| / |x| at its maximum (word 3).  Round-to-nearest errors are about
// uniform, so |dx| ~ 0.45 * 2^-8 |x| on ordinary data and the margin is ~2.3 x tighter than kHiHiRelErr - and exactly as wide
// as it must be on the data of test_worst_case_bf16_rounding.  The float32 accumulation of <= 384 exact products adds
// < 2.5e-5 |x||q|.  Returned in the ranking units of `l2` (2 x.q - |x|^2) / cosine / inner product.
__device__ __forceinline__ float hihi_margin(bool cosine, bool l2, float qn, float eq, const float *__restrict__ stats) {
    const float slop = 3.0e-5f;
    float m = cosine ? stats[3] * qn + (1.0f + stats[3]) * eq + slop * qn
                     : stats[2] * qn + (stats[0] + stats[2]) * eq + slop * stats[0] * qn;
    m *= (1.0f + 1e-5f) * MIR_MARGIN_SCALE;
    return l2 ? 2.0f * m : m;
}

// hihi_margin's slop covers the float32 accumulation of <= 384 products; the wide float32 sieve (d <= 1024) adds the rest:
// d 2^-24 of |x||q| (each of the d partial sums rounds once, relative 2^-24, and is at most |x||q|), with a little room
__host__ __device__ inline float wide_accum_slop(int d_pad) { return 1.1f * (float)d_pad * 5.9604645e-8f; }

// The same bound PER ROW, for the metrics that rank in the rows' own units (inner product, squared L2): with r = the index's
// largest |dx| / |x| (statistics word 3), |dx| <= r |x| for every row, so |x.q - hx.hq| <= |x| (r |q| + (1 + r) |dq| + slop |q|)
// = |x| * hihi_coeff(): the margin of a row is its norm times a per-query coefficient.  (Round 3 used the index's largest
// norm and largest residual for every row - DESIGN.md 7 gap 4, ADVICE r3: one long row among unit rows widened every row's
// margin, the candidate buffers overflowed and whole batches went to the exact pass.)  The sieve's filter takes a TILE's
// largest norm (one float per 32 rows, riding with the tile's norm column), its select each row's own.
__device__ __forceinline__ float hihi_coeff(bool l2, float qn, float eq, const float *__restrict__ stats) {
    const float slop = 3.0e-5f;
    float c = stats[3] * qn + (1.0f + stats[3]) * eq + slop * qn;
    c *= (1.0f + 2e-5f) * MIR_MARGIN_SCALE;  // (+ the float32 rounding of the norm it multiplies)
    return l2 ? 2.0f * c : c;
}
// one float per tile: the largest row norm of its 32 rows (row_dnorm_kernel's norms), rounded up; min_bits: the smallest of
// them over the index, as unsigned float bits (norm statistics word 4)
__global__ __launch_bounds__(256) void tile_maxnorm_kernel(const float *__restrict__ dnorm, int64_t n, uint32_t n_tiles,
                                                           float *__restrict__ tile_max, unsigned int *__restrict__ min_bits) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    float m = 0.f;
    for (int r = 0; r < kTileRows; ++r) {
        const int64_t row = (int64_t)t * kTileRows + r;
        if (row < n) {
            const float x = dnorm[row];
            m = (x > m || x != x) ? x : m;  // (a NaN norm stays: the tile's bound becomes NaN and everything in it passes)
        }
    }
    tile_max[t] = m * (1.0f + 1e-6f);
    if (m == m) atomicMin(min_bits, __float_as_uint(m));  // (non-negative floats order as their bits)
}
// f32 [n][d] row-major -> layout16.  One thread per (tile, block of a half, lane); ks32*32 >= d; columns past d and
// rows past n are 0.
// hi_only: a tile is its nb hi blocks alone (shards only the sieve scans), else nb hi blocks followed by nb lo blocks.
__global__ __launch_bounds__(256) void pack_split16_f32_kernel(const float *__restrict__ src, int64_t n, int d, int ks32,
                                                               int64_t total_lanes, uint4 *__restrict__ dst, bool hi_only = false) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_lanes) return;
    const int lane = (int)(gid & 63);
    const int64_t blk = gid >> 6;
    const int nb = ks32 * 2;  // blocks per half tile
    const int b = (int)(blk % nb);
    const int64_t tile = blk / nb;
    const int s = b >> 1, rh = b & 1;
    const int64_t row = tile * kTileRows + 16 * rh + (lane & 15);
    const int col0 = 32 * s + 8 * (lane >> 4);
    float x[8];
    if (row < n && col0 + 8 <= d && (d & 3) == 0) {
        const float4 *p = reinterpret_cast<const float4 *>(src + row * (int64_t)d + col0);
        const float4 a = p[0], c = p[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        x[4] = c.x; x[5] = c.y; x[6] = c.z; x[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (row < n && col0 + j < d) ? src[row * (int64_t)d + col0 + j] : 0.f;
    }
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf16(x[j], hi[j], lo[j]);
    if (hi_only) {
        dst[(tile * nb + b) * 64 + lane] = pack8(hi);
        return;
    }
    dst[(tile * (2 * nb) + b) * 64 + lane] = pack8(hi);
    dst[(tile * (2 * nb) + nb + b) * 64 + lane] = pack8(lo);
}

// Blocks [0, ntiles16*ks32): B-operand fragments of query tile w (16 queries), k-step s: lane l = (query 16 w + (l & 15),
// columns 32 s + 8 (l >> 4) .. + 7), hi then lo.  Blocks [ntiles16*ks32, +b): per-query sum of squares and norm in
// float64.  The first blocks also zero the control words (see prep_queries_kernel).
__global__ __launch_bounds__(64) void prep_queries16_kernel(const double *__restrict__ q, int b, int d, int ks32, int ntiles16,
                                                            uint4 *__restrict__ qsplit, double *__restrict__ q_sq,
                                                            double *__restrict__ q_norm, unsigned long long *__restrict__ gthr,
                                                            int gthr_words, double *__restrict__ q_err = nullptr) {
    const int lane = threadIdx.x, blk = blockIdx.x;
    if (gthr && blk * 64 + lane < gthr_words) gthr[blk * 64 + lane] = 0;
    if (blk < ntiles16 * ks32) {
        const int s = blk % ks32, w = blk / ks32;
        const int qi = 16 * w + (lane & 15);
        const int col0 = 32 * s + 8 * (lane >> 4);
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (qi < b && col0 + j < d) ? (float)q[(int64_t)qi * d + col0 + j] : 0.f;
            split_bf16(x, hi[j], lo[j]);
        }
        qsplit[((int64_t)blk * 2 + 0) * 64 + lane] = pack8(hi);
        qsplit[((int64_t)blk * 2 + 1) * 64 + lane] = pack8(lo);
    } else {
        const int qi = blk - ntiles16 * ks32;
        if (qi >= b) return;
        double s = 0.0, e2 = 0.0;
        for (int j = lane; j < d; j += 64) {
            const double x = q[(int64_t)qi * d + j];
            s += x * x;
            const double r = x - (double)bf16_bits_to_float(bf16_rne_bits((float)x));  // what the hi fragment loses of this component
            e2 += r * r;
        }
        s = wave_sum(s);
        e2 = wave_sum(e2);
        if (lane == 0) {
            q_sq[qi] = s;
            q_norm[qi] = sqrt(s);
            if (q_err) q_err[qi] = sqrt(e2);
        }
    }
}

constexpr int kQ16Queries = 128;
constexpr int kQ16MaxList = 60;  // klist the kernel takes (a query's buffer holds at most 64 entries, one per lane)
// a query's candidate buffer: klist kept entries + room for appends between compactions
__host__ __device__ constexpr int q16_buffer(int klist) { return klist + 16 < 64 ? klist + 16 : 64; }
__host__ __device__ constexpr int q16_ring_stages(int klist) { return q16_buffer(klist) <= 40 ? 5 : 4; }  // 24 KiB per stage at d = 384
__host__ __device__ constexpr size_t q16_lds_bytes(int ks32, int klist) {
    return (size_t)q16_ring_stages(klist) * ks32 * 2 * 1024 + (size_t)q16_buffer(klist) * kQ16Queries * 8;
}

// 64-bit wave minimum and the lane that holds it (ties: lowest lane)
__device__ __forceinline__ void wave_min_u64(uint64_t v, int lane, uint64_t &mn, int &pos) {
    uint64_t m = v;
    int p = lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(m >> 32), off, 64) << 32) | (uint64_t)__shfl_xor((uint32_t)m, off, 64);
        const int op = __shfl_xor(p, off, 64);
        if (o < m || (o == m && op < p)) { m = o; p = op; }
    }
    mn = m;
    pos = p;
}

// ---- shared by the 16-queries-per-wave kernels (this file and vec_kernels_h16.h): a wave's 16 candidate buffers ----
// One buffer per QUERY in LDS (`cap` entries, query-major behind `mylist`); the four lanes that hold a query's column
// (lane & 15) keep its threshold `thr` and fill count `cnt` replicated.
struct Q16Lists {
    uint64_t *mylist;  // this wave's 16 buffers
    int cap, klist, lane, qc, jg;
    unsigned long long colmask;  // the four lanes of this lane's column

    // Column c's buffer -> its best klist entries, in order; the klist-th becomes the column's threshold.  Whole wave.
    __device__ __forceinline__ void compact(int c, int &cnt, uint64_t &thr) const {
        uint64_t *lq = mylist + (size_t)c * cap;
        const int n = __builtin_amdgcn_readlane(cnt, c);  // (lane c is the column's first lane)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t mine = lane < n ? lq[lane] : 0;
        int rank = 0;
        for (int i0 = 0; i0 < n; i0 += 8) {  // eight broadcast reads in flight per step
            uint64_t o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = i0 + u < n ? lq[i0 + u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (o[u] > mine) ? 1 : 0;  // keys are distinct (they carry the row)
        }
        if (lane < n && rank < klist) lq[rank] = mine;  // every read above was issued before any of these writes
        const int kept = n < klist ? n : klist;
        uint64_t nthr = 0;
        if (n >= klist) {
            const unsigned long long at = __ballot(lane < n && rank == klist - 1);
            const int src = __builtin_ctzll(at);
            nthr = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), src) << 32) |
                   (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, src);
        }
        if (qc == c) {
            cnt = kept;
            if (nthr > thr) thr = nthr;
        }
    }

    // Candidates: bit r of `pm` = this lane's value v[r] (row row0 + 16 (r >> 2) + (r & 3)) passes.  Every lane appends its
    // lowest passing value per round; a column's up to four appends get consecutive slots from a wave vote; a buffer that
    // would overflow is compacted first.
    __device__ __forceinline__ void append(uint32_t pm, const float (&v)[8], uint32_t row0, int &cnt, uint64_t &thr) const {
        while (__any(pm != 0)) {
            const int r = pm ? __builtin_ctz(pm) : 0;
            // v[r] for a lane-dependent r, kept in registers (a plain select chain becomes an indexed scratch array whose
            // `s_waitcnt vmcnt(0)` would drain the DMA ring: the empty asm makes each element opaque)
            float x = v[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                float c = v[j];
                asm volatile("" : "+v"(c));
                x = (r == j) ? c : x;
            }
            x = (x == x) ? x + 0.0f : -__builtin_inff();  // NaN ranks last; -0 -> +0
            const uint64_t key = make_key(x, row0 + 16 * (r >> 2) + (r & 3));
            const bool ok = pm != 0 && key > thr;  // thr = 0 while the column has no threshold yet
            const unsigned long long bal = __ballot(ok);
            const int tot = __popcll(bal & colmask);
            if (__any(cnt + tot > cap)) {  // some column's buffer would overflow: compact those first, then redo the round
                unsigned long long over = __ballot(cnt + tot > cap && jg == 0);
                while (over) {
                    const int c = __builtin_ctzll(over);
                    over &= over - 1;
                    compact(c, cnt, thr);
                }
                continue;
            }
            if (ok) mylist[(size_t)qc * cap + cnt + __popcll(bal & colmask & ((1ull << lane) - 1ull))] = key;
            cnt += tot;
            pm &= pm - 1;
        }
    }

    // every buffer compacted once more (sorted, best first) and written out, empty entries as 0: [16][klist] for this wave
    __device__ __forceinline__ void write_out(uint64_t *out, int &cnt, uint64_t &thr) const {
        for (int c = 0; c < 16; ++c) {
            compact(c, cnt, thr);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int n = __builtin_amdgcn_readlane(cnt, c);
            if (lane < klist) out[(size_t)c * klist + lane] = lane < n ? mylist[(size_t)c * cap + lane] : 0;
        }
    }
};

// a tile's 32 norm values -> this lane's 8 (rows 16 rh + 4 jg + i): scalar fetch (a vector load would queue behind the
// DMA ring), then three selects per value
__device__ __forceinline__ void q16_load_aux(const float *__restrict__ aux, uint32_t t, int jg, float (&ax)[8]) {
    typedef uint32_t __attribute__((ext_vector_type(16))) u32x16;
    u32x16 sa, sb;
    const float *ap = aux + (size_t)__builtin_amdgcn_readfirstlane(t) * kTileRows;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(sa), "=&s"(sb)
                 : "s"(ap)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = __uint_as_float(sa[i]), a1 = __uint_as_float(sa[4 + i]), a2 = __uint_as_float(sa[8 + i]), a3 = __uint_as_float(sa[12 + i]);
        const float b0 = __uint_as_float(sb[i]), b1 = __uint_as_float(sb[4 + i]), b2 = __uint_as_float(sb[8 + i]), b3 = __uint_as_float(sb[12 + i]);
        ax[i] = jg == 0 ? a0 : jg == 1 ? a1 : jg == 2 ? a2 : a3;
        ax[4 + i] = jg == 0 ? b0 : jg == 1 ? b1 : jg == 2 ? b2 : b3;
    }
}

template <int KS32, int KIND, bool SAMPLE, int NS>
__global__ __launch_bounds__(512, 2) void scan_topk_q16_kernel(const uint4 *__restrict__ docs, const float *__restrict__ aux,
                                                               const uint4 *__restrict__ qsplit, const double *__restrict__ q_norm,
                                                               const float *__restrict__ max_norm, uint32_t n_rows,
                                                               uint32_t tile0, uint32_t n_tiles, int nq, int klist,
                                                               uint64_t *__restrict__ part, const uint64_t *__restrict__ gthr) {
    // this launch walks tiles [tile0, tile0 + n_tiles) of the shard
    constexpr int SB = KS32 * 2;          // 1-KiB blocks per stage
    constexpr int STAGE_U4 = SB * 64;
    constexpr int TILE_U4 = 2 * STAGE_U4;
    constexpr int PPW = SB / 8;           // DMA pieces per wave per stage
    constexpr int D = NS - 2;             // stages ahead: in stage 2t+1 slots {2t, 2t+1} are read and D more are in flight
    static_assert(SB % 8 == 0, "q16 scan: d padded to a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem);                                  // [NS][STAGE_U4]
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem + (size_t)NS * STAGE_U4 * 16);  // [128][cap]

    const int tid = threadIdx.x, lane = tid & 63, qc = lane & 15, jg = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qloc = wave8 * 16 + qc;              // this lane's query in the launch
    const bool lane_live = qloc < nq;
    const bool active = nq > wave8 * 16;
    const uint32_t G = gridDim.x;
    const int cap = q16_buffer(klist);
    uint64_t *mylist = lists + (size_t)wave8 * 16 * cap;  // this wave's 16 buffers, query-major
    const unsigned long long colmask = 0x0001000100010001ull << qc;  // the four lanes of this lane's column
    // per-query state, replicated in the four lanes of a column
    const uint64_t seed_thr = (SAMPLE || !lane_live) ? 0 : gthr[qloc];
    uint64_t thr = seed_thr;   // admission threshold: max(seed, list minimum once the list is full)
    int cnt = 0;               // entries in the query's list
    float best = -__builtin_inff();
    // margin of the hi*hi-only value, in the units the lists rank by
    float mg = 0.f;
    if (lane_live) {
        const float qn = (float)q_norm[qloc] * (1.0f + 1e-6f);
        mg = kHiHiRelErr * qn * (KIND == SCAN_COS ? 1.0f : max_norm[0]) * (KIND == SCAN_L2 ? 2.0f : 1.0f);
    }

    bf16x8 qh[KS32], ql[KS32];
    {
        const uint4 *qs = qsplit + (size_t)wave8 * KS32 * 128 + lane;
#pragma unroll
        for (int s = 0; s < KS32; ++s) {
            qh[s] = __builtin_bit_cast(bf16x8, qs[(s * 2 + 0) * 64]);
            ql[s] = __builtin_bit_cast(bf16x8, qs[(s * 2 + 1) * 64]);
        }
    }
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + G - 1) / G : 0;
    const uint32_t NG = my_tiles * 2;

    auto issue = [&](uint32_t g) {
        const uint32_t tile = tile0 + blockIdx.x + (g >> 1) * G;
        const uint4 *src = docs + (size_t)tile * TILE_U4 + (size_t)(g & 1) * STAGE_U4 + (wave8 * PPW) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(ring) + ((g % NS) * STAGE_U4 + (wave8 * PPW) * 64) * 16);
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_b128(src + i * 64, dst + i * 1024);
    };
    // ordinary loads are complete before the first DMA (the counted waits below count DMAs only)
#pragma unroll
    for (int s = 0; s < KS32; ++s) asm volatile("" : "+v"(qh[s]), "+v"(ql[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t g = 0; g < (uint32_t)D && g < NG; ++g) issue(g);

    auto wait_stage = [&](uint32_t g) {  // stage g has landed: all but the younger stages' pieces are done
        const uint32_t younger = (NG - 1 - g) < (uint32_t)(D - 1) ? (NG - 1 - g) : (uint32_t)(D - 1);
        if (younger == (uint32_t)(D - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // only in the last D - 1 stages of the launch
    };

    // ranking values of this lane's 8 rows from the dot products
    auto to_values = [&](const f32x4 &c0, const f32x4 &c1, const float (&ax)[8], float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = KIND == SCAN_IP ? c0[i] : KIND == SCAN_L2 ? fmaf(2.0f, c0[i], -ax[i]) : c0[i] * ax[i];
            v[4 + i] = KIND == SCAN_IP ? c1[i] : KIND == SCAN_L2 ? fmaf(2.0f, c1[i], -ax[4 + i]) : c1[i] * ax[4 + i];
        }
    };

    const Q16Lists L{mylist, cap, klist, lane, qc, jg, colmask};

    for (uint32_t ts = 0; ts < my_tiles; ++ts) {
        const uint32_t t = tile0 + blockIdx.x + ts * G;
        const uint32_t g = 2 * ts;
        const uint32_t row0 = t * kTileRows + 4 * jg;  // this lane's rows: row0 + 16 rh + i
        // ------------------------------------------------ stage 2t: the hi blocks, hi*hi for all of K
        wait_stage(g);
        __builtin_amdgcn_s_barrier();
        if (g + D < NG) issue(g + D);
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        float ax[8] = {};
        bool need0 = false, need1 = false;  // per 16-row half of the tile: may a row reach one of this wave's thresholds?
        if (active) {
            if (KIND != SCAN_IP) q16_load_aux(aux, t, jg, ax);
            const uint4 *st = ring + (size_t)(g % NS) * STAGE_U4 + lane;
            uint4 f0[3], f1[3];
            f0[0] = st[0 * 64]; f1[0] = st[1 * 64];
            f0[1] = st[2 * 64]; f1[1] = st[3 * 64];
#pragma unroll
            for (int s = 0; s < KS32; ++s) {
                if (s + 2 < KS32) {
                    f0[(s + 2) % 3] = st[(2 * (s + 2) + 0) * 64];
                    f1[(s + 2) % 3] = st[(2 * (s + 2) + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0[s % 3]), qh[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1[s % 3]), qh[s], c1, 0, 0, 0);
            }
            float v[8];
            to_values(c0, c1, ax, v);
            const float mx0 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), mx1 = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
            if (SAMPLE) {
                // a LOWER bound of this lane's best value: hi*hi minus the margin (sample tiles are whole tiles)
                if (lane_live) best = fmaxf(best, fmaxf(mx0, mx1) - mg);
            } else {
                // can any row of a half reach this lane's threshold?  (NaN passes: `!(x < y)`; an open list takes everything)
                const float vmin = thr == 0 ? -__builtin_inff() : key_value(thr);
                need0 = __any(lane_live && !(mx0 + mg < vmin));
                need1 = __any(lane_live && !(mx1 + mg < vmin));
            }
        }
        // ------------------------------------------------ stage 2t + 1: the lo blocks; corrections where needed
        wait_stage(g + 1);
        __builtin_amdgcn_s_barrier();
        if (g + 1 + D < NG) issue(g + 1 + D);
        if (!SAMPLE && (need0 || need1)) {
            // hi*lo and lo*hi for one 16-row half: two independent chains, both fragment streams three reads ahead
            auto correct = [&](int rh, f32x4 &c) {
                const uint4 *sh = ring + (size_t)(g % NS) * STAGE_U4 + rh * 64 + lane;
                const uint4 *sl = ring + (size_t)((g + 1) % NS) * STAGE_U4 + rh * 64 + lane;
                f32x4 e = {0.f, 0.f, 0.f, 0.f};
                uint4 fh[4], fl[4];
#pragma unroll
                for (int i = 0; i < 3; ++i) { fh[i] = sh[(2 * i) * 64]; fl[i] = sl[(2 * i) * 64]; }
#pragma unroll
                for (int s = 0; s < KS32; ++s) {
                    if (s + 3 < KS32) { fh[(s + 3) & 3] = sh[(2 * (s + 3)) * 64]; fl[(s + 3) & 3] = sl[(2 * (s + 3)) * 64]; }
                    __builtin_amdgcn_sched_barrier(0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fh[s & 3]), ql[s], c, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[s & 3]), qh[s], e, 0, 0, 0);
                }
                c += e;
            };
            if (need0) correct(0, c0);
            if (need1) correct(1, c1);
            float v[8];
            to_values(c0, c1, ax, v);
            const float vmin = thr == 0 ? -__builtin_inff() : key_value(thr);
            const uint32_t half_ok = (need0 ? 0x0fu : 0u) | (need1 ? 0xf0u : 0u);  // an uncorrected half holds hi*hi values only
            uint32_t pm = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t row = row0 + 16 * (r >> 2) + (r & 3);
                pm |= (uint32_t)(lane_live && row < n_rows && !(v[r] < vmin)) << r;
            }
            pm &= half_ok;
            L.append(pm, v, row0, cnt, thr);
        }
    }
    if (SAMPLE) {
        // four lanes hold a query's column: two values per query, each the maximum over distinct rows
        const float o = __shfl_xor(best, 16, 64);
        const float b2 = fmaxf(best, o);
        if (lane_live && (jg == 0 || jg == 2))
            reinterpret_cast<float *>(part)[((size_t)blockIdx.x * kQ16Queries + qloc) * 2 + (jg >> 1)] = b2;
        return;
    }
    // ---- every buffer compacted once more (sorted, best first) and written out, empty entries as 0: [128][klist] per workgroup ----
    L.write_out(part + (size_t)blockIdx.x * kQ16Queries * klist + (size_t)wave8 * 16 * klist, cnt, thr);
}

// Between the two launches of a progressive scan: a query's klist-th best key over the per-workgroup lists of the first
// launch is the klist-th best of the rows scanned so far - a subset of the shard, hence a valid (and, after 1/16 of the
// rows, tight: ~4.0 sigma at 10M rows against the sample pre-pass's 3.2) starting threshold for the rest.  One block per query.
__global__ __launch_bounds__(256) void list_threshold_kernel(const uint64_t *__restrict__ part, int nwg, int qpw, int klist, int nq,
                                                             unsigned long long *__restrict__ gthr) {
    __shared__ uint64_t keys[kMaxList];
    __shared__ uint64_t red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (q >= nq) return;
    merge_sorted_lists(part + (size_t)q * klist, (size_t)qpw * klist, nwg, klist, keys, red, tid);
    if (tid == 0) {
        const uint64_t kth = keys[klist - 1];
        if (kth != 0 && kth > gthr[q]) gthr[q] = kth;
    }
}

}  // namespace mir
