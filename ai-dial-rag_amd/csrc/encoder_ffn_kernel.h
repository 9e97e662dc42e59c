// The FFN kernel of encoder_ffn.hip, as a header so that diagnostic builds (tools/ffn_stamps.hip) can instantiate it
// with in-kernel time stamps; the library instantiates ffn_ln_kernel<false>.
#pragma once
#include "encoder_common.h"

namespace mir {
namespace enc {

// LDS fragment prefetch: a wave keeps FFN_RING - 1 ds_read_b128 in flight ahead of the MFMA that consumes them (measured
// with 4, 6, 8 and 12: no difference - fragment latency is not what bounds this kernel, see encoder_ffn.hip)
#ifndef FFN_RING
#define FFN_RING 4
#endif
// residual blocks the epilogue keeps in flight ahead of the one it is adding (measured 2 / 3 / 4 / 6: 873 / 882 / 883 / 904 us
// per 12288-tile pass - every block more costs registers the accumulators do not leave: 10 / 16 / 17 spills)
#ifndef FFN_EPI_AHEAD
#define FFN_EPI_AHEAD 2
#endif
#define FFN_NT ENC_NT  // non-temporal activation loads / stores (encoder_common.h)
// Of the six 1-KiB pieces of a W1 half that belong to token tile tl, the A wave of the tile moves the first
// FFN_A_PIECES and its B wave the rest (plus its six pieces of the W2 half): an LDS-DMA piece costs the issuing wave
// 75-100 cycles, so the split balances the two roles' stage times (tools/ffn_stamps.hip).
#ifndef FFN_A_PIECES
#define FFN_A_PIECES 2
#endif
constexpr int FFN_HBUF_BYTES = 2 * 4 * 2 * 64 * 16;  // h hand-off: [2 slots][4 tiles][2 fragments][64 lanes] x 16 B
constexpr int FFN_LUT_BYTES = 12 * 1024;            // the GELU table first: its offsets then fit a ds_read's 16-bit immediate
constexpr int FFN_MAX_GRID = 256;                   // one workgroup per CU of an MI355X
constexpr int FFN_LDS_BYTES = FFN_LUT_BYTES + 2 * FFN_STAGE_BYTES + FFN_HBUF_BYTES + FFN_PARAM_FLOATS * 4;
static_assert(GELU_LUT_FLOATS * 4 <= FFN_LUT_BYTES, "GELU table");

// The B waves' epilogue: y (+ bias) + residual -> LayerNorm -> ACT store - residual_ln_store's arithmetic in the same order
// (sums per half of the feature blocks, sequential over the blocks; v_cvt_pk_f16_f32 for the conversion), written for a
// wave whose registers are full: it runs INSIDE the persistent loop with all 192 accumulator registers live, and with the
// shared helper (parameters a block ahead, residual three blocks at a time, the scheduler free to overlap blocks) hipcc
// spilled most of y around it (180-270 registers; 17 when the same code ran once after the loop).  Here: the residual two
// blocks ahead in a ring of three (24 registers; FFN_EPI_AHEAD), parameters (LDS) a float4 at a time, a scheduling fence per block.
__device__ __forceinline__ void ffn_epilogue(f32x16 (&y)[NFB], const uint4 *__restrict__ resid_tile, const float *bias,
                                             const float *gamma, const float *beta, uint4 *__restrict__ out_tile, int lane,
                                             bool store) {
    const int h = lane >> 5;
    constexpr int HB = NFB / 2;
    // activations are read once and written once: non-temporal, so that they do not push the layer's weights (which every
    // workgroup re-reads) out of the L2s
    auto ld = [&](int i) {
#if FFN_NT
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(resid_tile) + i * 64 + lane);
        return make_uint4(t.x, t.y, t.z, t.w);
#else
        return resid_tile[i * 64 + lane];
#endif
    };
    auto st4 = [&](int i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if FFN_NT
        const u32x4 t = {a, b, c, d};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(out_tile) + i * 64 + lane);
#else
        out_tile[i * 64 + lane] = make_uint4(a, b, c, d);
#endif
    };
    constexpr int AHEAD = FFN_EPI_AHEAD;  // residual blocks in flight ahead of the one being added (ring of AHEAD + 1)
    uint4 rr[AHEAD + 1][2];
#pragma unroll
    for (int f = 0; f < AHEAD; ++f) {
        rr[f][0] = ld(f * 2 + 0);
        rr[f][1] = ld(f * 2 + 1);
    }
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NFB; ++f) {
        if (f + AHEAD < NFB) {
            rr[(f + AHEAD) % (AHEAD + 1)][0] = ld((f + AHEAD) * 2 + 0);
            rr[(f + AHEAD) % (AHEAD + 1)][1] = ld((f + AHEAD) * 2 + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float rv[8];
            frag_to_floats(rr[f % (AHEAD + 1)][s2], rv);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int g = 2 * s2 + gq;
                const float4 b4 = *reinterpret_cast<const float4 *>(bias + 32 * f + 8 * g + 4 * h);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = y[f][4 * g + i] + bb[i] + rv[4 * gq + i];
                    y[f][4 * g + i] = v;
                    sum[f / HB] += v;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const float mean = half_sum(sum[0] + sum[1]) * (1.0f / H);  // half A + half B
    float sq[2] = {0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NFB; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dlt = y[f][r] - mean;
            y[f][r] = dlt;
            sq[f / HB] = fmaf(dlt, dlt, sq[f / HB]);
        }
    const float rstd = rsqrtf(half_sum(sq[0] + sq[1]) * (1.0f / H) + LN_EPS);
#pragma unroll
    for (int f = 0; f < NFB; ++f) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint32_t w[4];
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int g = 2 * s2 + gq;
                const float4 g4 = *reinterpret_cast<const float4 *>(gamma + 32 * f + 8 * g + 4 * h);
                const float4 e4 = *reinterpret_cast<const float4 *>(beta + 32 * f + 8 * g + 4 * h);
                const float o0 = fmaf(y[f][4 * g + 0] * rstd, g4.x, e4.x), o1 = fmaf(y[f][4 * g + 1] * rstd, g4.y, e4.y);
                const float o2 = fmaf(y[f][4 * g + 2] * rstd, g4.z, e4.z), o3 = fmaf(y[f][4 * g + 3] * rstd, g4.w, e4.w);
                w[2 * gq] = pack2_rn(o0, o1);
                w[2 * gq + 1] = pack2_rn(o2, o3);
            }
            if (store) st4(f * 2 + s2, w[0], w[1], w[2], w[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// PERSISTENT: one workgroup per CU walks the 128-token groups blockIdx.x, blockIdx.x + gridDim.x, ... and the pipeline does
// not drain between them.  With one launch-time group per workgroup, a fifth of a workgroup's 81 us was prologue and
// epilogue with the matrix pipe idle (parameter + GELU table copy, x from HBM, W1(0..1) through registers; at the end the
// B waves' residual loads, LayerNorm and stores while the A waves had already left).  Across a group boundary now:
//   * the weight ring simply continues, W1 index modulo NHT: stages NHT-2 and NHT-1 bring W1(0), W1(1) of the NEXT group;
//   * the A waves request the next group's x during stage NHT-2, fragment ks as soon as its last MFMA has issued, and stage
//     NHT-1 (which used to have no product) computes the next group's first product;
//   * the B waves run the finished group's LayerNorm epilogue INSIDE stage 0 of the next group (where they have no product),
//     beside the A waves' first full stage.
// STAMPS (diagnostic builds only, tools/ffn_stamps.hip): workgroup 0 records s_memtime at the top of every stage (after
// the barrier) and at the end of its stage work, per wave, for its first group: stamps[wave][stage][2].
template <bool STAMPS>
__global__ __launch_bounds__(512, 2) void ffn_ln_kernel(const uint4 *__restrict__ act_in, int n_tiles,
                                                        const unsigned char *__restrict__ wffn,
                                                        const float *__restrict__ params, uint4 *__restrict__ act_out,
                                                        unsigned long long *__restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const uint32_t lut_base = __builtin_amdgcn_readfirstlane(enc_lds_addr(lds_all));
    unsigned char *smem = lds_all + FFN_LUT_BYTES;  // the two weight stages
    uint4 *hb = reinterpret_cast<uint4 *>(smem + 2 * FFN_STAGE_BYTES);
    float *prm = reinterpret_cast<float *>(smem + 2 * FFN_STAGE_BYTES + FFN_HBUF_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_b = wave >= 4;
    const int tl = wave & 3;  // token tile of the group
    const int n_groups = (n_tiles + 3) >> 2;
    const int g0 = blockIdx.x;
    // idle waves (a last group with fewer than four tiles) shadow a real tile: they must join barriers and the staging
    auto tile_of = [&](int g) { const int t = g * 4 + tl; return t < n_tiles ? t : n_tiles - 1; };
    const uint32_t lane16 = (uint32_t)lane * 16u;

    enc_stagger_start();
    for (int i = tid; i < FFN_PARAM_FLOATS; i += 512) prm[i] = params[i];
    for (int i = tid; i < GELU_LUT_FLOATS; i += 512) reinterpret_cast<float *>(lds_all)[i] = params[FFN_PARAM_FLOATS + i];

    if (!role_b) {
        // ------------------------------------------------------------------ role A
#ifdef FFN_A_PRIO
        __builtin_amdgcn_s_setprio(FFN_A_PRIO);
#endif
        const uint4 *xin = act_in + (size_t)tile_of(g0) * (NFB * 2 * 64) + lane;
        uint4 x[KS_H];
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) x[ks] = xin[ks * 64];
        // prologue staging through registers: W1(0) -> slot 1 and W1(1) -> slot 0, pieces 6 tl .. 6 tl + 5 per wave (from stage 1
        // on the B waves move the weights by LDS-DMA)
        {
            const size_t piece0 = (size_t)(tl * 6) * 64 + lane;
            const uint4 *s0 = reinterpret_cast<const uint4 *>(wffn) + piece0;                       // W1(0) = flat half 0
            const uint4 *s1 = reinterpret_cast<const uint4 *>(wffn + 2 * FFN_HALF_BYTES) + piece0;  // W1(1) = flat half 2
            uint4 *d1 = reinterpret_cast<uint4 *>(smem + FFN_STAGE_BYTES) + piece0, *d0 = reinterpret_cast<uint4 *>(smem) + piece0;
            uint4 t0[6], t1[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { t0[i] = s0[i * 64]; t1[i] = s1[i * 64]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) { d1[i * 64] = t0[i]; d0[i * 64] = t1[i]; }
        }
        __syncthreads();  // params, W1(0), W1(1) are in LDS

        // first product of tile 0, from stage -1 (slot 1, first half); the accumulator starts from the bias
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = prm[fi(r, h)];
        {
            const uint4 *st = reinterpret_cast<const uint4 *>(smem + FFN_STAGE_BYTES) + lane;
            uint4 fr[FFN_RING];
#pragma unroll
            for (int i = 0; i < FFN_RING - 1; ++i) fr[i] = st[i * 64];
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) {
                if (ks + FFN_RING - 1 < KS_H) fr[(ks + FFN_RING - 1) % FFN_RING] = st[(ks + FFN_RING - 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                hacc = mfma(fr[ks % FFN_RING], x[ks], hacc);
            }
        }
        // One stage of role A, in the order the instructions are to issue (a sched_barrier after every k-step keeps it): hipcc
        // left alone sinks the fragment reads next to their MFMAs (two reads in flight, each pair's LDS latency exposed:
        // 92 cycles per MFMA measured) and bunches the GELU at the end of the stage.  Per k-step: the fragment read
        // FFN_RING - 1 steps ahead, the MFMA of the NEXT tile's product, and a slice of the GELU of THIS tile - steps 0..15
        // look up one value each, even steps 6..20 finish a pair, steps 14 and 22 hand the two fragments over.
        // MODE 0: a stage inside a group.  MODE 1: stage NHT - 2, which also requests the next group's x (x_next: wave-uniform
        // base of that tile): fragment ks right after the last MFMA that reads x[ks].  As inline asm - a plain load of
        // read-only memory sinks to its first use, a stage later (encoder_kernels.h, oproj_ln_kernel) - so the registers are
        // pending until MODE 2's s_waitcnt + pass-through, and nothing may touch them in between.  MODE 2: stage NHT - 1, whose
        // "next product" is the next group's FIRST (bias of tile 0, W1(0) from the continuing ring, the new x).
        auto stage_a = [&](int s, f32x16 &cur, f32x16 &nxt, auto mode_, const uint4 *x_next) {
            constexpr int MODE = decltype(mode_)::value;
            const uint4 *st = reinterpret_cast<const uint4 *>(smem + (size_t)(s & 1) * FFN_STAGE_BYTES) + lane;  // W1(s+1)
            uint4 *ho = hb + ((size_t)((s & 1) * 4 + tl) * 2) * 64 + lane;
            uint4 fr[FFN_RING];
            if (MODE == 2) {
                asm volatile("s_waitcnt vmcnt(0) ; release-pending" ::: "memory");  // the x requested a stage ago (tools/check_pending_loads.py)
#pragma unroll
                for (int ks = 0; ks < KS_H; ++ks) {
                    u32x4 t = {x[ks].x, x[ks].y, x[ks].z, x[ks].w};
                    asm volatile("" : "+v"(t));
                    x[ks] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
            {
                const float *b1 = prm + 32 * (MODE == 2 ? 0 : s + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) nxt[r] = b1[fi(r, h)];  // the accumulator starts from the bias
#pragma unroll
                for (int i = 0; i < FFN_RING - 1; ++i) fr[i] = st[i * 64];
            }
            uint32_t addr[16], hw[8];
            float xc[16];
            f32x2 e[16];
            // this wave's FFN_A_PIECES 1-KiB pieces of W1(s + 2) (modulo NHT: the last two stages bring the next group's first
            // two tiles) -> first half of slot (s + 1) & 1, one every other k-step (the B waves move the rest and W2(s))
            const int w1n = s + 2 < NHT ? s + 2 : s + 2 - NHT;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(enc_lds_addr(smem) + (uint32_t)(((s + 1) & 1) * FFN_STAGE_BYTES + (tl * 6) * 1024));
            const uint4 *src = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * w1n) * FFN_HALF_BYTES) + (size_t)(tl * 6) * 64;  // wave-uniform
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) {
#ifndef FFN_ABL_NO_DMA
                if ((ks & 1) && (ks >> 1) < FFN_A_PIECES) enc_glds16_s(src + (ks >> 1) * 64, lane16, dst + (ks >> 1) * 1024);
#endif
                if (ks + FFN_RING - 1 < KS_H) fr[(ks + FFN_RING - 1) % FFN_RING] = st[(ks + FFN_RING - 1) * 64];
                nxt = mfma(fr[ks % FFN_RING], x[ks], nxt);
                if (MODE == 1) {
                    u32x4 t;
                    #if FFN_NT
                    asm volatile("global_load_dwordx4 %0, %1, %2 nt ; pending" : "=v"(t) : "v"(lane16), "s"(x_next + ks * 64) : "memory");
#else
                    asm volatile("global_load_dwordx4 %0, %1, %2 ; pending" : "=v"(t) : "v"(lane16), "s"(x_next + ks * 64) : "memory");
#endif
                    x[ks] = make_uint4(t.x, t.y, t.z, t.w);
                }
                if (ks < 16) {
                    addr[ks] = gelu_lut_addr(cur[ks], xc[ks], lut_base);
                    e[ks] = lds_read_f2(addr[ks]);
                }
                if (ks >= 6 && ks <= 20 && (ks & 1) == 0) {
                    const int p = (ks - 6) >> 1;
                    hw[p] = gelu_pack2(cur[2 * p], xc[2 * p], e[2 * p], cur[2 * p + 1], xc[2 * p + 1], e[2 * p + 1]);
                }
                if (ks == 14) ho[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                if (ks == 22) ho[64] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // this wave's pieces have landed before the barrier publishes them.  MODE 1: the x fragments requested after the
            // last piece (k-step 2 FFN_A_PIECES - 1) are younger and stay in flight.
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS_H - (2 * FFN_A_PIECES - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        f32x16 hnx;
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        for (int g = g0; g < n_groups; g += gridDim.x) {
            const bool st_on = STAMPS && blockIdx.x == 0 && lane == 0 && g == g0;
            const int gn = g + (int)gridDim.x < n_groups ? g + (int)gridDim.x : g;  // no next group: the prefetches re-read this one
            const uint4 *x_next = act_in + (size_t)tile_of(gn) * (NFB * 2 * 64);
            for (int s = 0; s < NHT - 2; s += 2) {  // two stages per trip: hacc / hnx swap roles instead of being copied (NHT is even)
                __syncthreads();  // stage s is complete in slot s & 1; everyone is done with slot (s + 1) & 1; h(s - 1) is visible
                if (st_on) stamps[(wave * (NHT + 1) + s) * 2] = __builtin_amdgcn_s_memtime();
                stage_a(s, hacc, hnx, M0{}, x_next);
                if (st_on) stamps[(wave * (NHT + 1) + s) * 2 + 1] = __builtin_amdgcn_s_memtime();
                __syncthreads();
                if (st_on) stamps[(wave * (NHT + 1) + s + 1) * 2] = __builtin_amdgcn_s_memtime();
                stage_a(s + 1, hnx, hacc, M0{}, x_next);
                if (st_on) stamps[(wave * (NHT + 1) + s + 1) * 2 + 1] = __builtin_amdgcn_s_memtime();
            }
            // the last two stages in straight-line code: between the x requests of the first and the s_waitcnt of the second
            // there must be no control-flow merge (a phi of pending registers becomes a copy of stale data)
            __syncthreads();
            if (st_on) stamps[(wave * (NHT + 1) + NHT - 2) * 2] = __builtin_amdgcn_s_memtime();
            stage_a(NHT - 2, hacc, hnx, M1{}, x_next);
            if (st_on) stamps[(wave * (NHT + 1) + NHT - 2) * 2 + 1] = __builtin_amdgcn_s_memtime();
            __syncthreads();
            if (st_on) stamps[(wave * (NHT + 1) + NHT - 1) * 2] = __builtin_amdgcn_s_memtime();
            stage_a(NHT - 1, hnx, hacc, M2{}, x_next);
            if (st_on) stamps[(wave * (NHT + 1) + NHT - 1) * 2 + 1] = __builtin_amdgcn_s_memtime();
            __syncthreads();  // stage NHT: role B's last product of the group
            if (st_on) {
                stamps[(wave * (NHT + 1) + NHT) * 2] = __builtin_amdgcn_s_memtime();
                stamps[(wave * (NHT + 1) + NHT) * 2 + 1] = __builtin_amdgcn_s_memtime();
            }
        }
    } else {
        // ------------------------------------------------------------------ role B
#ifdef FFN_B_PRIO
        __builtin_amdgcn_s_setprio(FFN_B_PRIO);
#endif
        f32x16 y[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) y[fb] = f32x16{0};
        __syncthreads();
        int tt_done = 0;           // the finished group's tile of this wave, its LayerNorm still to do
        bool live_done = false;
        // stage s+1 = [ W1(s+2) | W2(s) ] -> slot (s+1)&1 (W1 index modulo NHT: the next group's).  This wave moves pieces
        // 6 tl + FFN_A_PIECES .. 6 tl + 5 of the W1 half (the A wave of the tile moves the first FFN_A_PIECES) and pieces
        // 6 tl .. 6 tl + 5 of the W2 half; piece(s, i), i = 0 .. 11 - FFN_A_PIECES.  Nothing at s = NHT: the next group's stage 0
        // continues the ring.
        auto piece = [&](int s, int i) {
#ifndef FFN_ABL_NO_DMA  // (diagnostic builds: tools/ffn_stamps.hip)
            const int j = s + 1;
            const int w1n = j + 1 < NHT ? j + 1 : j + 1 - NHT;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(enc_lds_addr(smem) + (uint32_t)((j & 1) * FFN_STAGE_BYTES + (tl * 6) * 1024));
            // (wave-uniform bases in SGPRs + one lane offset: as per-lane 64-bit pointers these were hoisted out of the loops and spilled)
            const uint4 *src1 = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * w1n) * FFN_HALF_BYTES) + (size_t)(tl * 6) * 64;
            const uint4 *src2 = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * (j - 1) + 1) * FFN_HALF_BYTES) + (size_t)(tl * 6) * 64;
            if (s < NHT) {
                if (i < 6) enc_glds16_s(src2 + i * 64, lane16, dst + FFN_HALF_BYTES + i * 1024);
                else if (i < 12 - FFN_A_PIECES) enc_glds16_s(src1 + (i - 6 + FFN_A_PIECES) * 64, lane16, dst + (i - 6 + FFN_A_PIECES) * 1024);
            }
#endif
        };
        // One group.  Stage 0 has no product: the weight pieces of stage 1 and - unless this is the workgroup's FIRST group (a
        // compile-time flag: as a run-time condition around the epilogue hipcc spilled y at the top of every group) - the
        // finished group's epilogue, beside the A waves' first stage of this group.
        auto group_b = [&](auto first_, bool st_on) {
            constexpr bool FIRST = decltype(first_)::value;
            __syncthreads();
            if (st_on) stamps[(wave * (NHT + 1) + 0) * 2] = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 12; ++i) piece(0, i);
            if (!FIRST) {
                // (the lane index made opaque per group: otherwise the epilogue's 48 lane offsets are hoisted out of the group
                // loop, live through every stage, and spilled)
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                ffn_epilogue(y, act_in + (size_t)tt_done * (NFB * 2 * 64), prm + FF, prm + FF + H, prm + FF + 2 * H,
                             act_out + (size_t)tt_done * (NFB * 2 * 64), lane_e, live_done);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) y[fb] = f32x16{0};
            }
            // the pieces are older than the epilogue's 24 output stores (vmcnt counts stores on gfx9), which may stay in flight
            if (!FIRST && live_done) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (st_on) stamps[(wave * (NHT + 1) + 0) * 2 + 1] = __builtin_amdgcn_s_memtime();
            // stages 1 .. NHT: y += W2(s-1)^T h(s-1), one weight piece of stage s+1 per MFMA (issued in a burst at the top of the
            // stage they held this wave for ~100 cycles each before its first MFMA)
            for (int s = 1; s <= NHT; ++s) {
                __syncthreads();
                if (st_on) stamps[(wave * (NHT + 1) + s) * 2] = __builtin_amdgcn_s_memtime();
                const uint4 *st = reinterpret_cast<const uint4 *>(smem + (size_t)(s & 1) * FFN_STAGE_BYTES + FFN_HALF_BYTES) + lane;  // W2(s-1)
                const uint4 *hi = hb + ((size_t)(((s - 1) & 1) * 4 + tl) * 2) * 64 + lane;
                const uint4 h0 = hi[0], h1 = hi[64];
                uint4 fr[FFN_RING];
#pragma unroll
                for (int i = 0; i < FFN_RING - 1; ++i) fr[i] = st[i * 64];
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    if (i + FFN_RING - 1 < 24) fr[(i + FFN_RING - 1) % FFN_RING] = st[(i + FFN_RING - 1) * 64];
                    if (i < 12) piece(s, i);
                    __builtin_amdgcn_sched_barrier(0);
                    y[i >> 1] = mfma(fr[i % FFN_RING], (i & 1) ? h1 : h0, y[i >> 1]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of stage s+1 have landed before the barrier publishes them
                if (st_on) stamps[(wave * (NHT + 1) + s) * 2 + 1] = __builtin_amdgcn_s_memtime();
            }
        };
        group_b(std::true_type{}, STAMPS && blockIdx.x == 0 && lane == 0);
        tt_done = tile_of(g0);
        live_done = g0 * 4 + tl < n_tiles;
        for (int g = g0 + (int)gridDim.x; g < n_groups; g += gridDim.x) {
            group_b(std::false_type{}, false);
            tt_done = tile_of(g);
            live_done = g * 4 + tl < n_tiles;
        }
        ffn_epilogue(y, act_in + (size_t)tt_done * (NFB * 2 * 64), prm + FF, prm + FF + H, prm + FF + 2 * H,
                     act_out + (size_t)tt_done * (NFB * 2 * 64), lane, live_done);
    }
}


}  // namespace enc
}  // namespace mir
