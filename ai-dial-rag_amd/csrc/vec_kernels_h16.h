// The 128-query scan of the float16-NATIVE index (BASELINE config C5: d = 1024 float16, multimodal_retriever.py:96-153),
// round 2: 16 queries per wave over the full dimension on v_mfma_f32_16x16x32_f16, ONE product per fragment.
//
// Why: the earlier kernel (scan_topk_f16_kernel, 64 queries per pass, K-split over four waves, two MFMAs per k-step for
// the query's float16 hi and lo parts) ran at 68-69 % of the HBM roofline - power-limited like the float32 kernel it
// was modelled on.  The documents of this index are EXACT in float16, so the only rounding in a
// one-product scan is the query's: q_hi = f16(s q) with s a per-query power of two, |x.q - x.q_hi / s| <=
// 2^-11 |x||q| (+ the float32 accumulation).  That is useless as a result and fine for a FILTER: every value in the
// candidate lists is the hi-only value, the thresholds are in the same units, so the lists hold exactly the klist best
// rows BY THAT VALUE, and finalize_kernel's completeness check runs with the bound that belongs to it (kH16RelErr
// instead of scan_rel_err(d)): a query whose k-th and klist-th candidates are closer than the bound goes to the exact
// pass like any other unproven query.  (For float32 rows the same idea with bf16 - bound 1.7e-3 - flagged a query of
// nearly every batch; at 2^-11 the bound is a tenth of the gap between the 10th and the 18th best of 6.25M x 1024
// random rows.)  Half the MFMAs of the old kernel per byte, no correction pass, 128 queries per pass instead of 64.
//
// Index image: per 32-row tile, block (s, rh) = k-step of 32 columns s, row half rh at [2 s + rh]: 64 lanes x 8 f16,
// lane l = (row 16 rh + (l & 15), columns 32 s + 8 (l >> 4) .. + 7) - the A operand of v_mfma_f32_16x16x32_f16.
// A stage of the LDS-DMA ring is 16 k-steps = 32 blocks = 32 KiB (half of a d = 1024 tile).
// Candidate buffers (Q16Lists), thresholds, the sample pre-pass and the progressive two-launch scheme are those of
// vec_kernels_q16.h.
#pragma once
#include "vec_kernels_f16.h"
#include "vec_kernels_q16.h"

namespace mir {

// |x . q - (x . f16(s q)) / s| <= 2^-11 |x||q| for the rounding of the query (4.88e-4; components that round into
// float16's subnormal range add < 1e-8) plus the float32 accumulation of up to 1024 products (<= 1024 * 2^-24 = 6.1e-5 of
// sum |x_i q_i| <= |x||q|): 5.5e-4, rounded up.
constexpr double kH16RelErr = 6.0e-4;
#ifndef H16_STAGE_KS
#define H16_STAGE_KS 16
#endif
#ifndef H16_QT
#define H16_QT 1  // query tiles per wave of the float16 scan (see scan_topk_h16_kernel)
#endif
#ifndef H16_MAX_STAGES
#define H16_MAX_STAGES 4  // (launch_scan_h16 instantiates 2, 3 and 4)
#endif
static_assert(H16_MAX_STAGES <= 4, "add the instances to launch_scan_h16");
constexpr int kH16StageKs = H16_STAGE_KS;  // k-steps of 32 columns per ring stage (2 KiB each)
// The lists of this scan are k + kH16ListMargin long, half as much again as the float32 scans' margin: its bound is 30 x
// theirs, and with k + 8 about one random query in 10^4 on 6.25M x 1024 rows had its k-th and klist-th candidates closer
// than the bound and took the exact pass (10 ms there - ~6 % of the average throughput); the gap to the (k + 12)-th is 1.5 x
// as wide in the mean and the chance of falling under the bound ~1e-9 on the same data.  (k + 16 measured 3-5 % slower
// per step - longer lists to fill, write, merge and re-score - k + 12 about 2 %.)
constexpr int kH16ListMargin = 12;
// a query's candidate buffer: klist kept entries + room for 6 appended ones between compactions (k = 10: 28 entries; a
// four-stage ring leaves room for 32)
__host__ __device__ constexpr int h16_buffer(int klist) { return klist + 6 < 64 ? klist + 6 : 64; }
__host__ __device__ constexpr int h16_ring_stages(int klist) {  // 160 KiB of LDS: the most stages that fit beside the buffers
    int ns = H16_MAX_STAGES;
    while (ns > 2 && ns * kH16StageKs * 2048 + h16_buffer(klist) * kQ16Queries * 8 > 160 * 1024) --ns;
    return ns;
}
__host__ __device__ constexpr size_t h16_lds_bytes(int klist) {
    return (size_t)h16_ring_stages(klist) * kH16StageKs * 2048 + (size_t)h16_buffer(klist) * kQ16Queries * 8;
}

// f16 [n][d] row-major -> the image above.  One thread per (tile, block, lane); ks32 * 32 >= d; columns past d and rows
// past n are 0.
// The same image of a FLOAT32 matrix's bf16 hi parts (wide float32 shards: the sieve's filter, vec_kernels_sieve.h)
__global__ __launch_bounds__(256) void pack_hi16_f32_kernel(const float *__restrict__ src, int64_t n, int d, int ks32,
                                                            int64_t total_lanes, uint4 *__restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_lanes) return;
    const int lane = (int)(gid & 63);
    const int64_t blk = gid >> 6;
    const int nb = ks32 * 2;
    const int b = (int)(blk % nb);
    const int64_t tile = blk / nb;
    const int s = b >> 1, rh = b & 1;
    const int64_t row = tile * kTileRows + 16 * rh + (lane & 15);
    const int col0 = 32 * s + 8 * (lane >> 4);
    uint32_t hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t lo;
        split_bf16((row < n && col0 + j < d) ? src[row * (int64_t)d + col0 + j] : 0.f, hi[j], lo);
    }
    dst[blk * 64 + lane] = pack8(hi);
}

__global__ __launch_bounds__(256) void pack_f16_16_kernel(const _Float16 *__restrict__ src, int64_t n, int d, int ks32,
                                                          int64_t total_lanes, uint4 *__restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_lanes) return;
    const int lane = (int)(gid & 63);
    const int64_t blk = gid >> 6;
    const int nb = ks32 * 2;
    const int b = (int)(blk % nb);
    const int64_t tile = blk / nb;
    const int s = b >> 1, rh = b & 1;
    const int64_t row = tile * kTileRows + 16 * rh + (lane & 15);
    const int col0 = 32 * s + 8 * (lane >> 4);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n && col0 + 8 <= d && (d & 7) == 0) {
        v = *reinterpret_cast<const uint4 *>(src + row * (int64_t)d + col0);
    } else if (row < n) {
        uint32_t hbits[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            hbits[j] = col0 + j < d ? (uint32_t)__builtin_bit_cast(uint16_t, src[row * (int64_t)d + col0 + j]) : 0u;
        v = pack8(hbits);
    }
    dst[blk * 64 + lane] = v;
}

// One block per query: sum of squares and norm in float64, and 1 / s for the power of two s that puts max |q_i| into
// [128, 256) (s q is then a normal float16 number wherever q_i matters).  The first blocks also zero the control words
// (see prep_queries_kernel).
__global__ __launch_bounds__(64) void query_stats_h16_kernel(const double *__restrict__ q, int b, int d, double *__restrict__ q_sq,
                                                             double *__restrict__ q_norm, float *__restrict__ qscale_inv,
                                                             unsigned long long *__restrict__ gthr, int gthr_words) {
    const int lane = threadIdx.x, qi = blockIdx.x;
    if (gthr && qi * 64 + lane < gthr_words) gthr[qi * 64 + lane] = 0;
    if (qi >= b) return;
    double s = 0.0, m = 0.0;
    for (int j = lane; j < d; j += 64) {
        const double x = q[(int64_t)qi * d + j];
        s += x * x;
        const double ax = fabs(x);
        m = (ax == ax && ax > m) ? ax : m;
    }
    s = wave_sum(s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if (lane == 0) {
        q_sq[qi] = s;
        q_norm[qi] = sqrt(s);
        double sc = 1.0;
        if (m > 0.0 && m <= 1e300) {
            int e;
            frexp(m, &e);              // m = f * 2^e, f in [0.5, 1)
            sc = ldexp(1.0, 8 - e);    // s m in [128, 256)
        }
        qscale_inv[qi] = (float)(1.0 / sc);
    }
}

// Block (w, s): the B-operand fragment of query tile w (16 queries), k-step s: lane l = (query 16 w + (l & 15), columns
// 32 s + 8 (l >> 4) .. + 7), value f16(q / qscale_inv).  Runs after query_stats_h16_kernel.
__global__ __launch_bounds__(64) void prep_queries_h16_kernel(const double *__restrict__ q, int b, int d, int ks32,
                                                              const float *__restrict__ qscale_inv, uint4 *__restrict__ qfrag) {
    const int lane = threadIdx.x, blk = blockIdx.x;
    const int s = blk % ks32, w = blk / ks32;
    const int qi = 16 * w + (lane & 15);
    const int col0 = 32 * s + 8 * (lane >> 4);
    const double sc = qi < b ? 1.0 / (double)qscale_inv[qi] : 1.0;  // a power of two: exact
    uint32_t hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = (qi < b && col0 + j < d) ? (float)(q[(int64_t)qi * d + col0 + j] * sc) : 0.f;
        hi[j] = __builtin_bit_cast(uint16_t, (_Float16)x);
    }
    qfrag[(int64_t)blk * 64 + lane] = pack8(hi);
}

// QT = query tiles (16 queries each) per wave: 1 -> 8 waves, two per SIMD; 2 -> 4 waves, one per SIMD, each 1-KiB document
// fragment read from LDS once per wave feeds BOTH query tiles' MFMAs (half the LDS reads per byte streamed; the wave then
// holds 256 VGPRs of query fragments at d = 1024).
template <int KS32, int KIND, bool SAMPLE, int NS, int QT>
__global__ __launch_bounds__(512 / QT, QT == 1 ? 2 : 1) void scan_topk_h16_kernel(
    const uint4 *__restrict__ docs, const float *__restrict__ aux, const uint4 *__restrict__ qfrag, const float *__restrict__ qscale_inv,
    uint32_t n_rows, uint32_t tile0, uint32_t n_tiles, int nq, int klist, uint64_t *__restrict__ part, const uint64_t *__restrict__ gthr) {
    // this launch walks tiles [tile0, tile0 + n_tiles) of the shard
    static_assert(KS32 % kH16StageKs == 0, "h16 scan: d padded to a multiple of 512");
    static_assert(QT == 1 || QT == 2, "query tiles per wave");
    constexpr int WAVES = 8 / QT;
    constexpr int SPT = KS32 / kH16StageKs;   // stages per tile
    constexpr int SB = kH16StageKs * 2;       // 1-KiB blocks per stage
    constexpr int STAGE_U4 = SB * 64;
    constexpr int TILE_U4 = SPT * STAGE_U4;
    constexpr int PPW = SB / WAVES;           // DMA pieces per wave per stage
    constexpr int D = NS - 1;                 // stages in flight beyond the one being read
    static_assert(SB % WAVES == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem);                                      // [NS][STAGE_U4]
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem + (size_t)NS * STAGE_U4 * 16);  // [128][cap]

    const int tid = threadIdx.x, lane = tid & 63, qc = lane & 15, jg = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x;
    const int cap = h16_buffer(klist);
    const unsigned long long colmask = 0x0001000100010001ull << qc;  // the four lanes of this lane's column
    // per query tile u of the wave: the lane's query and its state (replicated in the four lanes of a column)
    int qloc[QT], cnt[QT];
    bool lane_live[QT], active[QT];
    uint64_t thr[QT];   // admission threshold: max(seed, list minimum once the list is full)
    float best[QT], inv_s[QT];
    f16x8 qh[QT][KS32];
    bool any_active = false;
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int t16 = wv * QT + u;  // the query tile's index in the launch
        qloc[u] = t16 * 16 + qc;
        lane_live[u] = qloc[u] < nq;
        active[u] = nq > t16 * 16;
        any_active |= active[u];
        thr[u] = (SAMPLE || !lane_live[u]) ? 0 : gthr[qloc[u]];
        cnt[u] = 0;
        best[u] = -__builtin_inff();
        inv_s[u] = lane_live[u] ? qscale_inv[qloc[u]] : 0.f;
        const uint4 *qs = qfrag + (size_t)t16 * KS32 * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS32; ++s) qh[u][s] = __builtin_bit_cast(f16x8, qs[s * 64]);
    }
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + G - 1) / G : 0;
    const uint32_t NG = my_tiles * SPT;

    auto issue = [&](uint32_t g) {
        const uint32_t tile = tile0 + blockIdx.x + (g / SPT) * G;
        const uint4 *src = docs + (size_t)tile * TILE_U4 + (size_t)(g % SPT) * STAGE_U4 + (wv * PPW) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(ring) + ((g % NS) * STAGE_U4 + (wv * PPW) * 64) * 16);
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_b128(src + i * 64, dst + i * 1024);
    };
    // ordinary loads are complete before the first DMA (the counted waits below count DMAs only)
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int s = 0; s < KS32; ++s) asm volatile("" : "+v"(qh[u][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t g = 0; g < (uint32_t)D && g < NG; ++g) issue(g);

    auto wait_stage = [&](uint32_t g) {  // stage g has landed: all but the younger stages' pieces are done
        const uint32_t younger = (NG - 1 - g) < (uint32_t)(D - 1) ? (NG - 1 - g) : (uint32_t)(D - 1);
        if (younger == (uint32_t)(D - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // only in the last D - 1 stages of the launch
    };

    // ranking values of this lane's 8 rows from the (scaled) dot products of query tile u
    auto to_values = [&](int u, const f32x4 &c0, const f32x4 &c1, const float (&ax)[8], float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d0 = c0[i] * inv_s[u], d1 = c1[i] * inv_s[u];
            v[i] = KIND == SCAN_IP ? d0 : KIND == SCAN_L2 ? fmaf(2.0f, d0, -ax[i]) : d0 * ax[i];
            v[4 + i] = KIND == SCAN_IP ? d1 : KIND == SCAN_L2 ? fmaf(2.0f, d1, -ax[4 + i]) : d1 * ax[4 + i];
        }
    };

    for (uint32_t ts = 0; ts < my_tiles; ++ts) {
        const uint32_t t = tile0 + blockIdx.x + ts * G;
        const uint32_t row0 = t * kTileRows + 4 * jg;  // this lane's rows: row0 + 16 rh + i
        f32x4 c0[QT], c1[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) { c0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; c1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        float ax[8] = {};
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const uint32_t g = ts * SPT + j;
            wait_stage(g);
            __builtin_amdgcn_s_barrier();
            if (g + D < NG) issue(g + D);
            if (any_active) {
                if (j == 0 && KIND != SCAN_IP) q16_load_aux(aux, t, jg, ax);
                const uint4 *st = ring + (size_t)(g % NS) * STAGE_U4 + lane;
                uint4 f0[3], f1[3];
                f0[0] = st[0 * 64]; f1[0] = st[1 * 64];
                f0[1] = st[2 * 64]; f1[1] = st[3 * 64];
#pragma unroll
                for (int s = 0; s < kH16StageKs; ++s) {
                    if (s + 2 < kH16StageKs) {
                        f0[(s + 2) % 3] = st[(2 * (s + 2) + 0) * 64];
                        f1[(s + 2) % 3] = st[(2 * (s + 2) + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < QT; ++u) {
                        c0[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f0[s % 3]), qh[u][j * kH16StageKs + s], c0[u], 0, 0, 0);
                        c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f1[s % 3]), qh[u][j * kH16StageKs + s], c1[u], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            if (!active[u]) continue;
            float v[8];
            to_values(u, c0[u], c1[u], ax, v);
            const float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            if (SAMPLE) {
                // thresholds and list values are in the same (hi-only) units: the maximum itself is the bound (sample tiles are whole tiles)
                if (lane_live[u]) best[u] = fmaxf(best[u], mx);
                continue;
            }
            // can any row reach this lane's threshold?  (NaN passes: `!(x < y)`; an open list takes everything)
            const float vmin0 = thr[u] == 0 ? -__builtin_inff() : key_value(thr[u]);
            if (!__any(lane_live[u] && !(mx < vmin0))) continue;
            uint32_t pm = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t row = row0 + 16 * (r >> 2) + (r & 3);
                pm |= (uint32_t)(lane_live[u] && row < n_rows && !(v[r] < vmin0)) << r;
            }
            const Q16Lists L{lists + (size_t)(wv * QT + u) * 16 * cap, cap, klist, lane, qc, jg, colmask};  // vec_kernels_q16.h
            L.append(pm, v, row0, cnt[u], thr[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        if (SAMPLE) {
            // four lanes hold a query's column: two values per query, each the maximum over distinct rows
            const float o = __shfl_xor(best[u], 16, 64);
            const float b2 = fmaxf(best[u], o);
            if (lane_live[u] && (jg == 0 || jg == 2))
                reinterpret_cast<float *>(part)[((size_t)blockIdx.x * kQ16Queries + qloc[u]) * 2 + (jg >> 1)] = b2;
        } else {
            // every buffer compacted once more (sorted, best first) and written out, empty entries as 0: [128][klist] per workgroup
            const Q16Lists L{lists + (size_t)(wv * QT + u) * 16 * cap, cap, klist, lane, qc, jg, colmask};
            L.write_out(part + (size_t)blockIdx.x * kQ16Queries * klist + (size_t)(wv * QT + u) * 16 * klist, cnt[u], thr[u]);
        }
    }
}

}  // namespace mir
