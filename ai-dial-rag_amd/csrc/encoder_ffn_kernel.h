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
// Of the six 1-KiB pieces of a W1 half that belong to token tile tl, the A wave of the tile moves the first
// FFN_A_PIECES and its B wave the rest (plus its six pieces of the W2 half): an LDS-DMA piece costs the issuing wave
// 75-100 cycles, so the split balances the two roles' stage times (tools/ffn_stamps.hip).
#ifndef FFN_A_PIECES
#define FFN_A_PIECES 2
#endif
constexpr int FFN_HBUF_BYTES = 2 * 4 * 2 * 64 * 16;  // h hand-off: [2 slots][4 tiles][2 fragments][64 lanes] x 16 B
constexpr int FFN_LUT_BYTES = 12 * 1024;            // the GELU table first: its offsets then fit a ds_read's 16-bit immediate
constexpr int FFN_LDS_BYTES = FFN_LUT_BYTES + 2 * FFN_STAGE_BYTES + FFN_HBUF_BYTES + FFN_PARAM_FLOATS * 4;
static_assert(GELU_LUT_FLOATS * 4 <= FFN_LUT_BYTES, "GELU table");

// STAMPS (diagnostic builds only, tools/ffn_stamps.hip): workgroup 0 records s_memtime at the top of every stage (after
// the barrier) and at the end of its stage work, per wave: stamps[wave][stage][2].
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
    const int tl = wave & 3;  // token tile of the workgroup
    const int tt_raw = blockIdx.x * 4 + tl;
    const bool live = tt_raw < n_tiles;
    const int tt = live ? tt_raw : n_tiles - 1;  // idle waves shadow a real tile: they must join barriers and the staging

    for (int i = tid; i < FFN_PARAM_FLOATS; i += 512) prm[i] = params[i];
    for (int i = tid; i < GELU_LUT_FLOATS; i += 512) reinterpret_cast<float *>(lds_all)[i] = params[FFN_PARAM_FLOATS + i];

    if (!role_b) {
        // ------------------------------------------------------------------ role A
#ifdef FFN_A_PRIO
        __builtin_amdgcn_s_setprio(FFN_A_PRIO);
#endif
        const uint4 *xin = act_in + (size_t)tt * (NFB * 2 * 64) + lane;
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
        auto stage_a = [&](int s, f32x16 &cur, f32x16 &nxt, auto with_next) {
            constexpr bool NEXT = decltype(with_next)::value;
            const uint4 *st = reinterpret_cast<const uint4 *>(smem + (size_t)(s & 1) * FFN_STAGE_BYTES) + lane;  // W1(s+1)
            uint4 *ho = hb + ((size_t)((s & 1) * 4 + tl) * 2) * 64 + lane;
            uint4 fr[FFN_RING];
            if (NEXT) {
                const float *b1 = prm + 32 * (s + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) nxt[r] = b1[fi(r, h)];  // the accumulator starts from the bias
#pragma unroll
                for (int i = 0; i < FFN_RING - 1; ++i) fr[i] = st[i * 64];
            }
            uint32_t addr[16], hw[8];
            float xc[16];
            f32x2 e[16];
            // this wave's six 1-KiB pieces of W1(s + 2) -> first half of slot (s + 1) & 1, one every other k-step (the B waves
            // move W2(s) into the second half)
            const bool dma = s + 2 < NHT;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(enc_lds_addr(smem) + (uint32_t)(((s + 1) & 1) * FFN_STAGE_BYTES + (tl * 6) * 1024));
            const uint4 *src = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * (s + 2)) * FFN_HALF_BYTES) + (size_t)(tl * 6) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) {
#ifndef FFN_ABL_NO_DMA
                if (dma && (ks & 1) && (ks >> 1) < FFN_A_PIECES) enc_glds16(src + (ks >> 1) * 64, dst + (ks >> 1) * 1024);
#endif
                if (NEXT) {
                    if (ks + FFN_RING - 1 < KS_H) fr[(ks + FFN_RING - 1) % FFN_RING] = st[(ks + FFN_RING - 1) * 64];
                    nxt = mfma(fr[ks % FFN_RING], x[ks], nxt);
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed before the barrier publishes them
        };
        f32x16 hnx;
        for (int s = 0; s < NHT; s += 2) {  // two stages per trip: hacc / hnx swap roles instead of being copied (NHT is even)
            __syncthreads();  // stage s is complete in slot s & 1; everyone is done with slot (s + 1) & 1; h(s - 1) is visible
            if (STAMPS && blockIdx.x == 0 && lane == 0) stamps[(wave * (NHT + 1) + s) * 2] = __builtin_amdgcn_s_memtime();
            stage_a(s, hacc, hnx, std::true_type{});
            if (STAMPS && blockIdx.x == 0 && lane == 0) stamps[(wave * (NHT + 1) + s) * 2 + 1] = __builtin_amdgcn_s_memtime();
            __syncthreads();
            if (STAMPS && blockIdx.x == 0 && lane == 0) stamps[(wave * (NHT + 1) + s + 1) * 2] = __builtin_amdgcn_s_memtime();
            if (s + 2 < NHT) stage_a(s + 1, hnx, hacc, std::true_type{});
            else stage_a(s + 1, hnx, hacc, std::false_type{});
            if (STAMPS && blockIdx.x == 0 && lane == 0) stamps[(wave * (NHT + 1) + s + 1) * 2 + 1] = __builtin_amdgcn_s_memtime();
        }
        __syncthreads();  // stage NHT: role B's last product
        if (STAMPS && blockIdx.x == 0 && lane == 0) {
            stamps[(wave * (NHT + 1) + NHT) * 2] = __builtin_amdgcn_s_memtime();
            stamps[(wave * (NHT + 1) + NHT) * 2 + 1] = __builtin_amdgcn_s_memtime();
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
        for (int s = 0; s <= NHT; ++s) {
            __syncthreads();
            if (STAMPS && blockIdx.x == 0 && lane == 0) stamps[(wave * (NHT + 1) + s) * 2] = __builtin_amdgcn_s_memtime();
            // stage s+1 = [ W1(s+2) | W2(s) ] -> slot (s+1)&1.  This wave moves pieces 6 tl + FFN_A_PIECES .. 6 tl + 5 of the W1 half
            // (the A wave of the tile moves the first FFN_A_PIECES) and pieces 6 tl .. 6 tl + 5 of the W2 half, ONE PER MFMA of
            // its product: issued in a burst at the top of the stage they held this wave for ~100 cycles each before its first MFMA.
            const int j = s + 1;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(enc_lds_addr(smem) + (uint32_t)((j & 1) * FFN_STAGE_BYTES + (tl * 6) * 1024));
            const uint4 *src1 = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * (j + 1)) * FFN_HALF_BYTES) + (size_t)(tl * 6) * 64 + lane;
            const uint4 *src2 = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * (j - 1) + 1) * FFN_HALF_BYTES) + (size_t)(tl * 6) * 64 + lane;
            const bool dma2 = s + 1 <= NHT, dma1 = dma2 && j + 1 < NHT;
            auto piece = [&](int i) {  // i = 0 .. 11 - FFN_A_PIECES
#ifndef FFN_ABL_NO_DMA  // (diagnostic builds: tools/ffn_stamps.hip)
                if (i < 6) { if (dma2) enc_glds16(src2 + i * 64, dst + FFN_HALF_BYTES + i * 1024); }
                else if (i < 12 - FFN_A_PIECES) { if (dma1) enc_glds16(src1 + (i - 6 + FFN_A_PIECES) * 64, dst + (i - 6 + FFN_A_PIECES) * 1024); }
#endif
            };
            if (s >= 1) {
                const uint4 *st = reinterpret_cast<const uint4 *>(smem + (size_t)(s & 1) * FFN_STAGE_BYTES + FFN_HALF_BYTES) + lane;  // W2(s-1)
                const uint4 *hi = hb + ((size_t)(((s - 1) & 1) * 4 + tl) * 2) * 64 + lane;
                const uint4 h0 = hi[0], h1 = hi[64];
                uint4 fr[FFN_RING];
#pragma unroll
                for (int i = 0; i < FFN_RING - 1; ++i) fr[i] = st[i * 64];
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    if (i + FFN_RING - 1 < 24) fr[(i + FFN_RING - 1) % FFN_RING] = st[(i + FFN_RING - 1) * 64];
                    if (i < 12) piece(i);
                    __builtin_amdgcn_sched_barrier(0);
                    y[i >> 1] = mfma(fr[i % FFN_RING], (i & 1) ? h1 : h0, y[i >> 1]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) piece(i);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of stage s+1 have landed before the barrier publishes them
            if (STAMPS && blockIdx.x == 0 && lane == 0) stamps[(wave * (NHT + 1) + s) * 2 + 1] = __builtin_amdgcn_s_memtime();
        }
        residual_ln_store<true, true>(y, act_in + (size_t)tt * (NFB * 2 * 64), prm + FF, prm + FF + H, prm + FF + 2 * H,
                          act_out + (size_t)tt * (NFB * 2 * 64), lane, live);
    }
}


}  // namespace enc
}  // namespace mir
