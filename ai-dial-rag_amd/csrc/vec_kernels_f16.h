// The 64-query K-split scan (round 1's float16-native kernel).  Since round 2 the float16-NATIVE index (BASELINE config
// C5) is scanned by vec_kernels_h16.h; what still runs here is the SPLIT = true instantiation: float32 rows with
// 384 < d <= 1024 (the multimodal / description retrievers' page embeddings) over their bf16 hi/lo image, three bf16
// MFMAs per k-step, K split over four waves per 32-query tile.  The SPLIT = false form (one 2-byte fragment stream, two
// f16 MFMAs per k-step for the query's hi and lo parts) is kept as the description of that layout:
// [tile of 32 rows][k-step][64 lanes][8 f16], lane l = row (l & 31), columns 16 ks + 8 (l >> 5) .. +7.
#pragma once
#include "vec_kernels.h"

namespace mir {

typedef _Float16 __attribute__((ext_vector_type(8))) f16x8;

// 64-query scan geometry
constexpr int kF16Queries = 64;
constexpr int kF16StageKsteps = 32;   // one stage = 32 k-steps = 32 KiB; each wave takes 8 of them
constexpr int kF16RingStages = 3;
constexpr int kF16Pending = 4;
__host__ __device__ constexpr size_t f16_lds_bytes(int klist) {
    return (size_t)kF16RingStages * kF16StageKsteps * 1024 + (size_t)(klist + kF16Pending) * 128 * 8 +
           2 * 3 * 16 * 64 * 4;
}

// 8 waves per workgroup, two per SIMD: query tile qt = wave >> 2 (32 queries each), k-slice
// j = wave & 3.  A stage is 32 consecutive k-steps of a doc tile (LDS-DMA ring as in
// the wide scans' ring: counted vmcnt, one raw s_barrier per stage); inside a stage wave (qt, j)
// multiplies k-steps 8j..8j+7 against ITS 8 query fragments of that stage (its quarter of the query
// tile's fragments stays in registers: 128 VGPRs at d = 1024), so every SIMD has work in every
// stage.  A tile's four partial 32x32 accumulators meet once per tile: three waves write theirs to
// LDS, the fourth (the query tile's reducer, j = 2 qt: on different SIMDs for the two query tiles)
// adds them, un-scales, applies the norm column and runs the candidate filter of
// lane-list filter one tile behind.
//
// SPLIT = true is the same kernel over the float32 index's bf16 hi/lo image (vec_kernels.h) for 384 < d <= 1024 -
// float32 page embeddings of the multimodal / description retrievers (embeddings_index.py:139-153 stores them as
// float32): a k-step is a (hi, lo) pair of 1-KiB blocks, three bf16 MFMAs per k-step (hi*hi, hi*lo, lo*hi), a
// stage is 16 k-steps (the same 32 KiB), the query is not scaled.  Before, such an index ran scan_topk_generic_kernel
// (32 queries per pass, fragments re-read from L2 per k-step, nothing in flight across tiles).
template <int KSTEPS, int KIND, bool SAMPLE, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void scan_topk_f16_kernel(const uint4 *__restrict__ docs,
                                                               const float *__restrict__ aux,
                                                               const uint4 *__restrict__ qfrag,
                                                               const float *__restrict__ qscale_inv, uint32_t n_rows,
                                                               uint32_t n_tiles, int nq, int klist,
                                                               uint64_t *__restrict__ part,
                                                               const uint64_t *__restrict__ gthr) {
    constexpr int BPK = SPLIT ? 2 : 1;        // 1-KiB blocks per k-step
    constexpr int SK = kF16StageKsteps / BPK; // k-steps per stage (32 KiB either way)
    static_assert(KSTEPS % SK == 0 && KSTEPS / SK >= 2, "wide scan: whole stages, at least two per tile");
    constexpr int SPT = KSTEPS / SK;          // stages per tile (>= 2: the exchange buffer is single)
    constexpr int WK = SK / 4;                // k-steps per wave per stage
    constexpr int QK = SPT * WK;              // k-steps per wave per tile
    static_assert(QK <= 16, "query fragments beyond 128 VGPRs per wave spill");
    constexpr int NS = kF16RingStages;
    constexpr int STAGE_U4 = SK * BPK * 64;   // 1 KiB per block
    constexpr int PPW = SK * BPK / 8;         // DMA pieces per wave per stage
    constexpr int TILE_U4 = KSTEPS * BPK * 64;
    constexpr int LS = 128;                   // list column stride
    typedef uint32_t __attribute__((ext_vector_type(16))) u32x16;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem);                                            // [NS][STAGE_U4]
    uint64_t *list = reinterpret_cast<uint64_t *>(smem + (size_t)NS * STAGE_U4 * 16);         // [klist + pending][128]
    float4 *xbuf = reinterpret_cast<float4 *>(list + (size_t)(klist + kF16Pending) * LS);     // [2][3][4][64]
    uint64_t *stage_out = reinterpret_cast<uint64_t *>(smem);                                 // [64][klist], reuses the ring

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, qj = lane & 31;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = wave8 >> 2, j = wave8 & 3;
    const bool is_red = j == 2 * qt;
    const int xslot = (j - 2 * qt - 1) & 3;   // 0..2 for the three non-reducers
    const int ltid = qt * 64 + lane;          // list column (reducers only)
    const bool active = nq > 32 * qt;
    const bool lane_live = qj + 32 * qt < nq;
    const uint32_t G = gridDim.x;

    if (is_red && !SAMPLE) {
        for (int p = 0; p < klist + kF16Pending; ++p) list[p * LS + ltid] = 0;
    }
    float best = -__builtin_inff();
    uint64_t minkey = 0;
    int minpos = 0, pending = 0;
    const uint64_t seed_thr = (!is_red || SAMPLE) ? 0 : gthr[qt * 32 + qj];
    const float qinv = SPLIT ? 1.0f : (is_red && lane_live) ? qscale_inv[32 * qt + qj] : 0.f;

    // this wave's quarter of its query tile's fragments: slot s <-> k-step (s / WK) * SK + j * WK + s % WK
    f16x8 qh[QK], ql[QK];
#pragma unroll
    for (int s = 0; s < QK; ++s) {
        const int kg = (s / WK) * SK + j * WK + (s % WK);
        const uint4 *qs = qfrag + ((size_t)qt * KSTEPS + kg) * 128 + lane;
        qh[s] = __builtin_bit_cast(f16x8, qs[0]);
        ql[s] = __builtin_bit_cast(f16x8, qs[64]);
    }
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + G - 1) / G : 0;
    const uint32_t NG = my_tiles * SPT;

    auto issue = [&](uint32_t g) {
        const uint32_t tile = blockIdx.x + (g / SPT) * G;
        const uint4 *src = docs + (size_t)tile * TILE_U4 + (size_t)(g % SPT) * STAGE_U4 + (wave8 * PPW) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(ring) + ((g % NS) * STAGE_U4 + (wave8 * PPW) * 64) * 16);
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_b128(src + i * 64, dst + i * 1024);
    };
    // ordinary loads are complete before the first DMA (the counted waits below count DMAs only)
#pragma unroll
    for (int s = 0; s < QK; ++s) asm volatile("" : "+v"(qh[s]), "+v"(ql[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t g = 0; g < (uint32_t)(NS - 1) && g < NG; ++g) issue(g);

    f32x16 pacc = {0};          // reducer: its own partial of the previous tile
    float4 pax[4] = {};         // reducer: the previous tile's norm values (sqeuclid / cosine)
    uint32_t prow0 = n_rows;    // "no previous tile"
    float4 *xb_w = xbuf + ((size_t)(qt * 3 + xslot) * 4) * 64 + lane;   // where a non-reducer writes
    const float4 *xb_r = xbuf + ((size_t)(qt * 3) * 4) * 64 + lane;     // where the reducer reads (3 slots x 4 chunks)

    auto load_aux = [&](uint32_t t, float4 (&ax)[4]) {
        u32x16 sa, sb;
        const float *ap = aux + (size_t)__builtin_amdgcn_readfirstlane(t) * kTileRows;
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(sa), "=&s"(sb)
                     : "s"(ap)
                     : "memory");
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float lo[4], hi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = 8 * gq + i;
                lo[i] = __uint_as_float(idx < 16 ? sa[idx] : sb[idx - 16]);
                hi[i] = __uint_as_float(idx + 4 < 16 ? sa[idx + 4] : sb[idx + 4 - 16]);
            }
            ax[gq] = h ? make_float4(hi[0], hi[1], hi[2], hi[3]) : make_float4(lo[0], lo[1], lo[2], lo[3]);
        }
    };

    // reducer: scores of the previous tile = (own partial + the three others') / s, then the metric's form
    auto epilogue = [&](const float4 (&w4)[4]) {
        float pv[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float wv[4] = {w4[c].x, w4[c].y, w4[c].z, w4[c].w};
            const float av[4] = {pax[c].x, pax[c].y, pax[c].z, pax[c].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dot = (pacc[4 * c + i] + wv[i]) * qinv;
                pv[4 * c + i] = KIND == SCAN_L2 ? fmaf(2.0f, dot, -av[i]) : KIND == SCAN_COS ? dot * av[i] : dot;
            }
        }
        const float m01 = fmaxf(fmaxf(pv[0], pv[1]), pv[2]), m02 = fmaxf(fmaxf(pv[3], pv[4]), pv[5]);
        const float m03 = fmaxf(fmaxf(pv[6], pv[7]), pv[8]), m04 = fmaxf(fmaxf(pv[9], pv[10]), pv[11]);
        const float m05 = fmaxf(fmaxf(pv[12], pv[13]), pv[14]);
        const float mx = fmaxf(fmaxf(fmaxf(m01, m02), fmaxf(m03, m04)), fmaxf(m05, pv[15]));
        if (SAMPLE) {
            if (prow0 != n_rows) best = fmaxf(best, mx);
            return;
        }
        const uint64_t other = ((uint64_t)__shfl_xor((uint32_t)(minkey >> 32), 32, 64) << 32) |
                               (uint64_t)__shfl_xor((uint32_t)minkey, 32, 64);
        uint64_t thr = other > minkey ? other : minkey;
        thr = seed_thr > thr ? seed_thr : thr;
        const float vmin = thr == 0 ? -__builtin_inff() : key_value(thr);
        if (!__any(!(mx < vmin) && prow0 != n_rows)) return;
        uint32_t pmask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) pmask |= (uint32_t)(!(pv[r] < vmin)) << r;
        const bool full_tile = __builtin_amdgcn_readfirstlane(prow0 != n_rows ? (prow0 & ~31u) + 32 <= n_rows : 0);
        if (!full_tile) {
            uint32_t ok = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) ok |= (uint32_t)(prow0 + 8 * (r >> 2) + (r & 3) < n_rows) << r;
            pmask &= ok;
        }
        pmask = prow0 == n_rows ? 0u : pmask;
        drain_candidates<LS>(pmask, pv, prow0, list, klist, ltid, minkey, minpos, pending);
    };

    uint32_t g = 0;
    for (uint32_t ts = 0; ts < my_tiles; ++ts) {
        const uint32_t t = blockIdx.x + ts * G;
        f32x16 acc = {0};
        float4 w4[4] = {};
#pragma unroll
        for (int part_i = 0; part_i < SPT; ++part_i, ++g) {
            const uint32_t younger = (NG - 1 - g) < (uint32_t)(NS - 2) ? (NG - 1 - g) : (uint32_t)(NS - 2);
            if (younger == NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage g is in LDS; stage g-1 is consumed; last tile's partials are written
            if (g + NS - 1 < NG) issue(g + NS - 1);
            if (part_i == 0 && is_red && active) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 a = xb_r[(0 * 4 + c) * 64], b = xb_r[(1 * 4 + c) * 64], d3 = xb_r[(2 * 4 + c) * 64];
                    w4[c] = make_float4(a.x + b.x + d3.x, a.y + b.y + d3.y, a.z + b.z + d3.z, a.w + b.w + d3.w);
                }
            }
            if (active) {
                const uint4 *st = ring + (size_t)(g % NS) * STAGE_U4 + (j * WK * BPK) * 64 + lane;
                if constexpr (!SPLIT) {
                    uint4 fr[3];
                    fr[0] = st[0 * 64];
                    fr[1] = st[1 * 64];
#pragma unroll
                    for (int i = 0; i < WK; ++i) {
                        if (i + 2 < WK) fr[(i + 2) % 3] = st[(i + 2) * 64];
                        __builtin_amdgcn_sched_barrier(0);
                        const f16x8 a = __builtin_bit_cast(f16x8, fr[i % 3]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qh[part_i * WK + i], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ql[part_i * WK + i], acc, 0, 0, 0);
                    }
                } else {
                    uint4 fh[3], fl[3];
                    fh[0] = st[0 * 64];
                    fl[0] = st[1 * 64];
                    fh[1] = st[2 * 64];
                    fl[1] = st[3 * 64];
#pragma unroll
                    for (int i = 0; i < WK; ++i) {
                        if (i + 2 < WK) {
                            fh[(i + 2) % 3] = st[(2 * (i + 2) + 0) * 64];
                            fl[(i + 2) % 3] = st[(2 * (i + 2) + 1) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, fh[i % 3]), al = __builtin_bit_cast(bf16x8, fl[i % 3]);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, qh[part_i * WK + i]), bl = __builtin_bit_cast(bf16x8, ql[part_i * WK + i]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    }
                }
                if (part_i == SPT - 1 && is_red) epilogue(w4);
            }
        }
        if (active) {
            if (!is_red) {  // read by the reducer after the next barrier
#pragma unroll
                for (int c = 0; c < 4; ++c) xb_w[c * 64] = make_float4(acc[4 * c + 0], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
            } else {
                if (KIND != SCAN_IP) load_aux(t, pax);
                pacc = acc;
                prow0 = lane_live ? t * kTileRows + 4 * h : n_rows;
            }
        }
    }
    // the last tile: its partials are complete after one more barrier
    __syncthreads();
    if (is_red && active) {
        float4 w4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a = xb_r[(0 * 4 + c) * 64], b = xb_r[(1 * 4 + c) * 64], d3 = xb_r[(2 * 4 + c) * 64];
            w4[c] = make_float4(a.x + b.x + d3.x, a.y + b.y + d3.y, a.z + b.z + d3.z, a.w + b.w + d3.w);
        }
        epilogue(w4);
        for (int i = 0; i < pending; ++i) {
            const uint64_t key = list[(klist + i) * LS + ltid];
            if (key > minkey) list_insert<LS>(list, klist, ltid, key, minkey, minpos);
        }
    }
    if (SAMPLE) {
        if (is_red) reinterpret_cast<float *>(part)[((size_t)blockIdx.x * kF16Queries + 32 * qt + qj) * 2 + h] = best;
        return;
    }

    // ---- merge the two half-lists of each query, write [64][klist] per workgroup ----
    __syncthreads();
    for (int i = tid; i < kF16Queries * klist; i += 512) stage_out[i] = 0;
    __syncthreads();
    if (is_red) {
        const int qloc = 32 * qt + qj;
        for (int p = 0; p < klist; ++p) {
            const uint64_t key = list[p * LS + ltid];
            if (key == 0) continue;
            int rank = 0;
            const int t2 = ltid ^ 32;
            for (int p2 = 0; p2 < klist; ++p2) {
                rank += (list[p2 * LS + ltid] > key) ? 1 : 0;
                rank += (list[p2 * LS + t2] > key) ? 1 : 0;
            }
            if (rank < klist) stage_out[qloc * klist + rank] = key;
        }
    }
    __syncthreads();
    uint64_t *out = part + (size_t)blockIdx.x * kF16Queries * klist;
    for (int i = tid; i < kF16Queries * klist; i += 512) out[i] = stage_out[i];
}

}  // namespace mir
