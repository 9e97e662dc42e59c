// Kernels of the bge-small-en encoder except attention (encoder_attention.hip); see encoder_common.h
// for the design notes, layouts and helpers.  Included by encoder.hip only (the kernels have external linkage).
#pragma once
#include "encoder_common.h"

namespace mir {
namespace enc {

// ---------------------------------------------------------------- E1: embeddings + LN
// one wave per token tile; block = 256 (4 tiles)
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t *__restrict__ ids, const TileInfo *__restrict__ ti,
                                                       int n_tiles, const float *__restrict__ word,
                                                       const float *__restrict__ pos, const float *__restrict__ type0,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       uint4 *__restrict__ act) {
    const int lane = threadIdx.x & 63, h = lane >> 5, t_in = lane & 31;
    const int tt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tt >= n_tiles) return;
    const TileInfo info = ti[tt];
    const int p = 32 * (tt - info.seq_first_tile) + t_in;
    const int id = ids[tt * 32 + t_in];
    const float *wrow = word + (size_t)id * H;
    const float *prow = pos + (size_t)(p < 512 ? p : 511) * H;
    float v[NFB][16];
    float sum = 0.f;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * fb + 8 * g + 4 * h;  // registers 4g..4g+3 are 4 consecutive features
            const float4 a = *reinterpret_cast<const float4 *>(wrow + f0);
            const float4 b = *reinterpret_cast<const float4 *>(prow + f0);
            const float4 c = *reinterpret_cast<const float4 *>(type0 + f0);
            v[fb][4 * g + 0] = a.x + b.x + c.x;
            v[fb][4 * g + 1] = a.y + b.y + c.y;
            v[fb][4 * g + 2] = a.z + b.z + c.z;
            v[fb][4 * g + 3] = a.w + b.w + c.w;
            sum += v[fb][4 * g] + v[fb][4 * g + 1] + v[fb][4 * g + 2] + v[fb][4 * g + 3];
        }
    const float mean = half_sum(sum) * (1.0f / H);
    float sq = 0.f;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dlt = v[fb][r] - mean;
            sq = fmaf(dlt, dlt, sq);
        }
    const float rstd = rsqrtf(half_sum(sq) * (1.0f / H) + LN_EPS);
    uint4 *out = act + (size_t)tt * (NFB * 2 * 64);
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * fb + fi(r, h);
            o[r] = fmaf((v[fb][r] - mean) * rstd, gamma[f], beta[f]);
        }
        out[(fb * 2 + 0) * 64 + lane] = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
        out[(fb * 2 + 1) * 64 + lane] = make_uint4(pack2(o[8], o[9]), pack2(o[10], o[11]), pack2(o[12], o[13]), pack2(o[14], o[15]));
    }
}

// ---------------------------------------------------------------- E2: QKV projection
// One wave per QKV_G = 2 token tiles (64 tokens), 8 waves per block = 512 tokens, two waves per SIMD.  wqkv: 36 tiles x
// 24 k-steps of 1-KiB fragments: tiles 0-11 = Q heads, 12-23 = K heads, 24-35 = V heads.  The weights enter LDS ONCE per
// workgroup: a ring of QKV_NS stages (one stage = one weight tile = 24 KiB), filled by LDS-DMA (3 pieces per wave and
// stage) with counted waits, one barrier per stage; a fragment read from LDS feeds two MFMAs (two token tiles'
// activations stay in registers: 192 VGPRs).  Round 1's form had every wave stream all 864 KiB through its own register
// ring - four copies per CU, at the ~29 B/clk a CU draws from L2 that stream (57 us per pass) was longer than the MFMAs.
// Measured per 12288-tile pass: three tiles per wave and one wave per SIMD (288 VGPRs of activations) 376-381 us, this
// form 363 us; letting the second wave of each SIMD run half a stage late (its epilogue under the first wave's MFMAs)
// needs the accumulators across the barrier, spills 28 registers, and every spill reload waits behind the DMAs in
// flight: 484 us.  By ablation the 376 us are ~200 us of MFMAs, ~75 us attributable to the Q/K/V stores (0.9 GB per pass)
// and the per-stage epilogue; the kernel is not L2-bound (24 KiB of weights per stage and CU).  Starting the workgroups
// in four phases 2.4 or 4.8 us apart (enc_stagger_start, which helps the persistent kernels) changes nothing here: 348 / 350 / 352 us.
// Q, K come out as W^T x^T (rows = head features, lanes = tokens); V as x W (rows = tokens, lanes = head features) so
// that each is directly the operand the attention kernel needs.  Output fragment buffers: [tile][head][s2][64].
#ifndef QKV_G_
#define QKV_G_ 2
#endif
#ifndef QKV_WAVES_
#define QKV_WAVES_ 8
#endif
constexpr int QKV_G = QKV_G_;                                 // token tiles per wave
constexpr int QKV_WAVES = QKV_WAVES_;                         // waves per workgroup
constexpr int QKV_PIECES = KS_H / QKV_WAVES;                  // DMA pieces per wave and stage
#ifndef QKV_NS_
#define QKV_NS_ 4
#endif
constexpr int QKV_NS = QKV_NS_;                               // ring stages
constexpr int QKV_STAGE_BYTES = KS_H * 1024;                  // one weight tile
constexpr int QKV_LDS_BYTES = QKV_NS * QKV_STAGE_BYTES + 3 * H * 4;  // + the biases

__global__ __launch_bounds__(64 * QKV_WAVES, 1) void qkv_kernel(const uint4 *__restrict__ act, int n_tiles,
                                                     const uint4 *__restrict__ wqkv, const float *__restrict__ bqkv,
                                                     uint4 *__restrict__ qf, uint4 *__restrict__ kf,
                                                     uint4 *__restrict__ vf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *bias = reinterpret_cast<float *>(smem + QKV_NS * QKV_STAGE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = (blockIdx.x * QKV_WAVES + wave) * QKV_G;
    int tt[QKV_G];
    bool live[QKV_G];
#pragma unroll
    for (int g = 0; g < QKV_G; ++g) {
        live[g] = t0 + g < n_tiles;
        tt[g] = live[g] ? t0 + g : n_tiles - 1;  // a missing tile shadows a real one; nothing is stored for it (idle waves still
    }                                             // join the barriers and move their share of the weights)
    for (int i = tid; i < 3 * H; i += 64 * QKV_WAVES) bias[i] = bqkv[i];
    __syncthreads();  // (the stage barriers below are raw s_barrier: they do not wait for LDS writes)
    uint4 x[QKV_G][KS_H];
#pragma unroll
    for (int g = 0; g < QKV_G; ++g) {
        const uint4 *xin = act + (size_t)tt[g] * (NFB * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) x[g][ks] = enc_load_nt(xin + ks * 64);  // read once here (the FFN's LayerNorm re-reads it as the residual much later)
    }
    // ordinary loads are complete before the first DMA: the counted waits below count DMAs and stores only
#pragma unroll
    for (int g = 0; g < QKV_G; ++g)
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) asm volatile("" : "+v"(x[g][ks].x), "+v"(x[g][ks].y), "+v"(x[g][ks].z), "+v"(x[g][ks].w));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    auto issue = [&](int tile) {  // weight tile `tile` -> slot tile % QKV_NS; this wave moves pieces 6 wave .. 6 wave + 5
        const uint4 *src = wqkv + (size_t)tile * (KS_H * 64) + (size_t)(wave * QKV_PIECES) * 64;  // wave-uniform
        const uint32_t dst = __builtin_amdgcn_readfirstlane(enc_lds_addr(smem) + (uint32_t)((tile % QKV_NS) * QKV_STAGE_BYTES + (wave * QKV_PIECES) * 1024));
#pragma unroll
        for (int i = 0; i < QKV_PIECES; ++i) enc_glds16_s(src + i * 64, (uint32_t)lane * 16u, dst + i * 1024);
    };
    for (int t = 0; t < QKV_NS - 1; ++t) issue(t);

    // Two loops with a compile-time operand order: with `is_v ? mfma(x, w) : mfma(w, x)` inside one loop hipcc emitted a
    // branch per MFMA.
    auto run_tiles = [&](int tile_lo, int tile_hi, auto IS_V_) {
        constexpr bool is_v = decltype(IS_V_)::value;
        for (int tile = tile_lo; tile < tile_hi; ++tile) {
            // Younger than this stage's DMA pieces, in issue order: stores(tile-3), DMA(tile+1), stores(tile-2), DMA(tile+2),
            // stores(tile-1) (for QKV_NS = 4; in general QKV_NS - 1 store batches and QKV_NS - 2 DMA batches) once the pipeline is full;
            // the first and last stages simply drain.
            if (tile >= QKV_NS - 1 && tile + QKV_NS - 1 <= 36) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((QKV_NS - 1) * 2 * QKV_G + (QKV_NS - 2) * QKV_PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage `tile` is in LDS for everyone; everyone is done with stage tile - 1
            if (tile + QKV_NS - 1 < 36) issue(tile + QKV_NS - 1);
            const uint4 *w = reinterpret_cast<const uint4 *>(smem + (size_t)(tile % QKV_NS) * QKV_STAGE_BYTES) + lane;
            f32x16 acc[QKV_G];
#pragma unroll
            for (int g = 0; g < QKV_G; ++g) acc[g] = f32x16{0};
            uint4 fr[4];
            fr[0] = w[0 * 64];
            fr[1] = w[1 * 64];
            fr[2] = w[2 * 64];
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) {
                if (ks + 3 < KS_H) fr[(ks + 3) & 3] = w[(ks + 3) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < QKV_G; ++g)
                    acc[g] = is_v ? mfma(x[g][ks], fr[ks & 3], acc[g]) : mfma(fr[ks & 3], x[g][ks], acc[g]);
            }
            const int head = tile % 12;
            const float *b = bias + tile * 32;
            uint4 *dbase = (tile < 12 ? qf : tile < 24 ? kf : vf);
#pragma unroll
            for (int g = 0; g < QKV_G; ++g) {
                if (is_v) {
                    const float bv = b[lane & 31];  // column = head feature
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[g][r] += bv;
                } else {
                    const float qs = tile < 12 ? kQScaleLog2e : 1.0f;  // Q carries the softmax scale (see attention_kernel)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[g][r] = (acc[g][r] + b[fi(r, h)]) * qs;
                }
                // always two stores per token tile (the counted waits assume it): a missing tile rewrites its shadow's values
                uint4 *dst = dbase + ((size_t)(tt[g] * NH + head) * 2) * 64 + lane;
                dst[0] = acc_to_frag(acc[g], 0);
                dst[64] = acc_to_frag(acc[g], 1);
            }
        }
    };
    run_tiles(0, 24, std::false_type{});
    run_tiles(24, 36, std::true_type{});
}

// ---------------------------------------------------------------- E3a: attention
// attention_kernel lives in encoder_attention.hip: a translation unit of its own, built with
// -mllvm -amdgpu-mfma-vgpr-form=1 so that its MFMA results land in VGPRs.  It is VALU-bound and
// touches every accumulator register with VALU instructions; with accumulators in AGPRs (hipcc's
// default) that cost a v_accvgpr_read/write per touch, ~130 of its ~700 instructions.  The same
// option makes oproj_ln_kernel spill (192 accumulators + 96 activations need the AGPR half), hence
// the separate file.
int32_t launch_attention2(const uint4 *qf, const uint4 *kf, const uint4 *vf, const int32_t *units, int n_units, uint4 *ctx,
                          hipStream_t stream);  // units: n_units x int4, see attention2_kernel
constexpr int kFqaTiles = 8;  // token tiles (= waves) per bin of the fused QKV + attention kernel (encoder_attention.hip)
int32_t fqa_prepare();        // once per process: dynamic-LDS attribute
int32_t launch_fused_qkv_attention(const uint4 *act, const uint4 *wqkv, const float *bqkv, const int32_t *bins, int n_bins, uint4 *ctx,
                                   hipStream_t stream);
int32_t launch_attention(const uint4 *qf, const uint4 *kf, const uint4 *vf, const TileInfo *ti, int n_tiles,
                         uint4 *ctx, hipStream_t stream);
// latency path, every sequence of the batch a single tile: QKV projection + attention in one dispatch (encoder_attention.hip)
int32_t launch_qkv_attention_single(const uint4 *act, const uint4 *wqkv, const float *bqkv, const TileInfo *ti, int n_tiles,
                                    uint4 *ctx, hipStream_t stream);
// ... with the previous layer's closing LayerNorm (Y + bias + residual -> act_out) in front
int32_t launch_ln_qkv_attention_single(const float *Y, const uint4 *resid, const float *bias, const float *gamma, const float *beta,
                                       uint4 *act_out, const uint4 *wqkv, const float *bqkv, const TileInfo *ti, int n_tiles,
                                       uint4 *ctx, hipStream_t stream);

// ---------------------------------------------------------------- E3b: output projection + residual + LN
// Workgroup = 8 waves = one GROUP of 128 tokens (4 token tiles) at a time, two waves per SIMD: wave (tile tl, half hf)
// accumulates the tile's output blocks 6 hf .. 6 hf + 5 (96 registers).  One workgroup per CU, PERSISTENT over the groups
// blockIdx.x, blockIdx.x + gridDim.x, ...: with one launch-time group per workgroup (round 2's first form: the tile's
// context held in 96 registers, 220 VGPRs, so ONE workgroup per CU) the phases of a group ran one after the other - wait
// for the context from HBM, six stages of products, wait for the residual, LayerNorm, stores - 23 us per group for 6 us
// of MFMAs, the matrix pipe 15 % busy.  Now the products run K-OUTER: stage st = k-steps 4 st .. 4 st + 3 of W_o for
// all 12 output blocks (48 KiB, two slots) and of the group's four context tiles (16 KiB, three slots), both by LDS-DMA,
// W_o one stage ahead (L2) and the context two stages ahead (HBM) - across the group boundary, so the next group's first
// context stages land during this group's LayerNorm; the residual fragments are requested at the start of the last stage.
// Every accumulator still sees its 24 products in ascending k, so the sums are the ones oproj_small_kernel forms.
// LayerNorm: the two waves of a tile exchange their partial sums through LDS (ln_part_* in encoder_common.h: the
// statistics are defined as half A + half B everywhere, so this kernel and the latency path round alike).
constexpr int OPROJ_KC = 4;                                 // k-steps per stage
constexpr int OPROJ_NST = KS_H / OPROJ_KC;                  // 6 stages per group
constexpr int OPROJ_W_SLOT = NFB * OPROJ_KC * 1024;         // [12 output blocks][4 k-steps] fragments of 1 KiB
constexpr int OPROJ_C_SLOT = 4 * OPROJ_KC * 1024;           // [4 token tiles][4 k-steps]
constexpr int OPROJ_C_BASE = 2 * OPROJ_W_SLOT;
constexpr int OPROJ_XS_BASE = OPROJ_C_BASE + 3 * OPROJ_C_SLOT;
constexpr int OPROJ_PAR_BASE = OPROJ_XS_BASE + 2 * 8 * 64 * 4;  // after the [2 rounds][8 waves][64 lanes] partial sums
constexpr int OPROJ_LDS_BYTES = OPROJ_PAR_BASE + 3 * H * 4;      // + bias | gamma | beta

constexpr int OPROJ_MAX_GRID = 256;                         // one workgroup per CU of an MI355X
static_assert(OPROJ_NST % 2 == 0 && OPROJ_NST % 3 == 0, "slot indices continue across groups");

__global__ __launch_bounds__(512, 1) void oproj_ln_kernel(const uint4 *__restrict__ ctx, int n_tiles,
                                                          const uint4 *__restrict__ wo, const float *__restrict__ bo,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const uint4 *__restrict__ act_in, uint4 *__restrict__ act_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem + OPROJ_XS_BASE);  // [2][8][64]
    constexpr int HB = NFB / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = wave & 3, hf = wave >> 2;
    const int n_groups = (n_tiles + 3) >> 2;
    const uint32_t lds0 = enc_lds_addr(smem);

    // W_o stage st -> slot st & 1; this wave moves pieces 6 wave .. 6 wave + 5 (piece p = block p / 4, k-step 4 st + p % 4)
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto issue_w = [&](int st) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((st & 1) * OPROJ_W_SLOT + wave * 6 * 1024));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = wave * 6 + i;
            enc_glds16_s(wo + (size_t)((p >> 2) * KS_H + OPROJ_KC * st + (p & 3)) * 64, lane16, dst + i * 1024);
        }
    };
    // context stage st of group g -> slot st % 3; this wave moves k-steps 2 (wave & 1), + 1 of the group's tile wave / 2
    auto issue_c = [&](int g, int st) {
        const int t_raw = g * 4 + (wave >> 1);
        const int t = t_raw < n_tiles ? t_raw : n_tiles - 1;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(OPROJ_C_BASE + (st % 3) * OPROJ_C_SLOT + wave * 2 * 1024));
        const uint4 *src = ctx + (size_t)t * (NFB * 2 * 64) + (size_t)(OPROJ_KC * st + 2 * (wave & 1)) * 64;
        enc_glds16_s_nt(src, lane16, dst);  // the context is read once
        enc_glds16_s_nt(src + 64, lane16, dst + 1024);
    };

    enc_stagger_start();
    int g = blockIdx.x;
    issue_w(0);
    issue_c(g, 0);
    issue_c(g, 1);
    // bias | gamma | beta (contiguous at bo; `gamma`, `beta` point into the same block) once per workgroup into LDS: every
    // wave reads all of its 288 parameter values per group, which as global loads were 72 KiB per wave and group through the
    // texture path - as much as the data (241 -> 212 us per 12288-tile pass)
    for (int i = tid; i < 3 * H; i += 512) reinterpret_cast<float *>(smem + OPROJ_PAR_BASE)[i] = bo[i];  // visible after the first stage's barrier
    bool stores_in_flight = false;
    for (; g < n_groups; g += gridDim.x) {
        const int gn = g + (int)gridDim.x < n_groups ? g + (int)gridDim.x : g;  // no next group: the prefetch re-reads this one
        const int tt_raw = g * 4 + tl;
        const bool live = tt_raw < n_tiles;
        const int tt = live ? tt_raw : n_tiles - 1;  // idle waves shadow a real tile: they join the barriers and the DMA
        const uint4 *resid = act_in + (size_t)tt * (NFB * 2 * 64);
        uint4 rr[HB * 2];
        f32x16 y[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) y[j] = f32x16{0};
#pragma unroll
        for (int st = 0; st < OPROJ_NST; ++st) {
            // this wave's pieces of stage st have landed; what may still be in flight is younger: the two context pieces of
            // stage st + 1 and, at a group's first stage, the previous group's 12 output stores (vmcnt counts stores on gfx9)
            if (st == 0 && stores_in_flight)
                asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else if (st == OPROJ_NST - 1)
                asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // + the 12 residual loads of the stage before
            else
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __syncthreads();  // stage st is complete; everyone is done with stage st - 1's slots
            issue_w(st + 1 < OPROJ_NST ? st + 1 : 0);
            if (st + 2 < OPROJ_NST) issue_c(g, st + 2); else issue_c(gn, st + 2 - OPROJ_NST);
            // the residual fragments, requested two stages before the LayerNorm.  As asm: hipcc moves a plain load of
            // read-only memory down to its first use (the loads are not on the memory chain, sched_barrier does not hold
            // them), which is after the last MFMA.  The registers must not be touched until the s_waitcnt + pass-through
            // below: tools/check_pending_loads.py reads the ISA of a build for exactly that.
            if (st == OPROJ_NST - 2) {
#pragma unroll
                for (int i = 0; i < HB * 2; ++i) {
                    u32x4 t;
                    #if ENC_NT
                    asm volatile("global_load_dwordx4 %0, %1, %2 nt ; pending" : "=v"(t) : "v"(lane16), "s"(resid + (HB * hf * 2 + i) * 64) : "memory");
#else
                    asm volatile("global_load_dwordx4 %0, %1, %2 ; pending" : "=v"(t) : "v"(lane16), "s"(resid + (HB * hf * 2 + i) * 64) : "memory");
#endif
                    rr[i] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
            const uint4 *wl = reinterpret_cast<const uint4 *>(smem + (size_t)(st & 1) * OPROJ_W_SLOT + (size_t)(HB * hf) * OPROJ_KC * 1024) + lane;
            const uint4 *cl = reinterpret_cast<const uint4 *>(smem + OPROJ_C_BASE + (size_t)(st % 3) * OPROJ_C_SLOT + (size_t)tl * OPROJ_KC * 1024) + lane;
            uint4 cf[2], wf[2][HB];
            cf[0] = cl[0];
#pragma unroll
            for (int j = 0; j < HB; ++j) wf[0][j] = wl[(j * OPROJ_KC) * 64];
#pragma unroll
            for (int kk = 0; kk < OPROJ_KC; ++kk) {
                if (kk + 1 < OPROJ_KC) {
                    cf[(kk + 1) & 1] = cl[(kk + 1) * 64];
#pragma unroll
                    for (int j = 0; j < HB; ++j) wf[(kk + 1) & 1][j] = wl[(j * OPROJ_KC + kk + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HB; ++j) y[j] = mfma(wf[kk & 1][j], cf[kk & 1], y[j]);
            }
        }
        // LayerNorm over the tile's 384 features: this wave holds blocks [6 hf, 6 hf + 6)
        // the table's address is made opaque per group: left loop-invariant, hipcc keeps a dozen of its lane addresses in
        // registers through the product stages and spills some (each reload in the store phase waits on vmcnt(0))
        uint32_t par_off = OPROJ_PAR_BASE;
        asm volatile("" : "+v"(par_off));
        const float *p_bias = reinterpret_cast<const float *>(smem + par_off), *p_gamma = p_bias + H, *p_beta = p_bias + 2 * H;
        asm volatile("s_waitcnt vmcnt(8) ; release-pending" ::: "memory");  // younger than the residual loads: the last stage's 6 + 2 DMA pieces
#pragma unroll
        for (int i = 0; i < HB * 2; ++i) {
            u32x4 t = {rr[i].x, rr[i].y, rr[i].z, rr[i].w};
            asm volatile("" : "+v"(t));
            rr[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
        const float ps = ln_part_sum_rr<HB>(y, HB * hf, rr, p_bias, lane);
        xs[(0 * 8 + wave) * 64 + lane] = ps;
        __syncthreads();
        const float mean = half_sum(xs[(0 * 8 + tl) * 64 + lane] + xs[(0 * 8 + 4 + tl) * 64 + lane]) * (1.0f / H);  // half A + half B
        const float pq = ln_part_sq<HB>(y, mean);
        xs[(1 * 8 + wave) * 64 + lane] = pq;
        __syncthreads();
        const float rstd = rsqrtf(half_sum(xs[(1 * 8 + tl) * 64 + lane] + xs[(1 * 8 + 4 + tl) * 64 + lane]) * (1.0f / H) + LN_EPS);
        ln_part_store<HB, false, true, true>(y, HB * hf, rstd, p_gamma, p_beta, act_out + (size_t)tt * (NFB * 2 * 64), lane, live);
        stores_in_flight = live;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last prefetches target this workgroup's LDS: they land before it is released
}

// ---------------------------------------------------------------- E4: FFN1 + GELU + FFN2 + residual + LN
// ffn_ln_kernel lives in encoder_ffn.hip (a translation unit of its own, MFMA accumulators in VGPRs: its two wave
// roles overlay their registers, which needs one register class).
int32_t launch_ffn(const uint4 *act_in, int n_tiles, const unsigned char *wffn, const float *params, uint4 *act_out,
                   hipStream_t stream);
int32_t ffn_prepare();  // once per process: dynamic-LDS attribute

// ---------------------------------------------------------------- L: latency path for tiny inputs
// The kernels above are throughput-shaped: a workgroup walks ALL weights of a layer for its 128
// tokens, so ONE query (a single 32-token tile; `aembed_query`, embeddings.py:93-96) took 2.1 ms with
// 255 CUs idle.  For at most kSmallTiles token tiles the same products are spread over the weight
// dimension instead: one WAVE per (weight tile, token tile), weights straight from L2, no LDS.
//   qkv_small      grid (36, tiles)   24 MFMAs per wave
//   oproj_small    grid (12, tiles)   24 MFMAs -> pre-LayerNorm float32 tile rows in Y
//   ffn1_small     grid (48, tiles)   24 MFMAs + GELU -> h fragments
//   ffn2_small     grid (12, tiles)   96 MFMAs over the h fragments -> Y
//   ln_small       grid (tiles)       Y + bias + residual -> LayerNorm -> ACT (four waves per tile)
//   ln_ffn1_small  grid (12, tiles)   the attention block's LayerNorm (repeated per workgroup, into LDS) + four FFN1 tiles
//   qkv_attention_single, ln_qkv_attention_single (encoder_attention.hip)  grid (12, tiles): QKV + attention of a head when
//                  every sequence is one tile; the second with the previous layer's closing LayerNorm in front
// Y: float32 [tile][12 output tiles][16 registers][64 lanes] (the accumulators as they are).
// Same arithmetic in the same order as the throughput kernels (MFMA chains over k ascending, the same
// GELU and residual_ln_store): a sequence's embedding is bit-identical on either path, which
// test_batching_is_invariant holds both to.
// measured crossover (tools/encoder_latency.py): 64 tiles 1.07 vs 2.38 ms, 256 tiles 1.80 vs 2.49 ms, 512 tiles 2.87 vs 2.62 ms
#ifndef ENC_SMALL_TILES
#define ENC_SMALL_TILES 256
#endif
constexpr int kSmallTiles = ENC_SMALL_TILES;
// one wave per SIMD is the plan for these kernels (a handful of waves per CU at most): tell the scheduler, or it
// keeps register pressure low for an occupancy nobody wants and sinks the prefetches back next to their uses
#define MIR_ONE_WAVE __attribute__((amdgpu_waves_per_eu(1, 1)))

__global__ __launch_bounds__(64) MIR_ONE_WAVE void qkv_small_kernel(const uint4 *__restrict__ act, const uint4 *__restrict__ wqkv,
                                                       const float *__restrict__ bqkv, uint4 *__restrict__ qf,
                                                       uint4 *__restrict__ kf, uint4 *__restrict__ vf) {
    const int lane = threadIdx.x, h = lane >> 5;
    const int tile = blockIdx.x, tt = blockIdx.y;
    const uint4 *xin = act + (size_t)tt * (NFB * 2 * 64) + lane;
    const uint4 *wp = wqkv + (size_t)tile * (KS_H * 64) + lane;
    uint4 x[KS_H], w[KS_H];
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) { x[ks] = xin[ks * 64]; w[ks] = wp[ks * 64]; }
    f32x16 acc = {0};
    const float *b = bqkv + tile * 32;
    if (tile >= 24) {  // V: x W (rows = tokens)
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) acc = mfma(x[ks], w[ks], acc);
        const float bv = b[lane & 31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += bv;
    } else {           // Q, K: W^T x^T (rows = head features)
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) acc = mfma(w[ks], x[ks], acc);
        const float qs = tile < 12 ? kQScaleLog2e : 1.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + b[fi(r, h)]) * qs;
    }
    uint4 *dst = (tile < 12 ? qf : tile < 24 ? kf : vf) + ((size_t)(tt * NH + tile % 12) * 2) * 64 + lane;
    dst[0] = acc_to_frag(acc, 0);
    dst[64] = acc_to_frag(acc, 1);
}

__device__ __forceinline__ void store_acc_rows(float *__restrict__ Y, int tt, int nt, int lane, const f32x16 &acc) {
    float4 *yo = reinterpret_cast<float4 *>(Y + (((size_t)tt * NFB + nt) * 16) * 64) + lane;  // [4 chunks][64 lanes] float4
#pragma unroll
    for (int c = 0; c < 4; ++c) yo[c * 64] = make_float4(acc[4 * c + 0], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
}

// Y + bias + residual -> LayerNorm -> ACT; FOUR waves per token tile, three feature blocks each.  The LayerNorm statistics
// are defined everywhere as half A + half B, each half a sequential float32 sum over its six blocks (ln_part_* in
// encoder_common.h), so waves 0 and 2 start the halves' sums and waves 1 and 3 CONTINUE them from the value their partner
// left in LDS: the same additions in the same order as the throughput kernels, bit for bit.  (One wave per tile loaded
// all 192 accumulator rows and took ~10 us, half of a single query's GPU time: 489 -> 377 us per query with two waves.
// Folding the LayerNorm into the product kernels - the tile's last wave to finish does it, counted with an atomic
// between two device-scope fences - was measured too: the fences write back / invalidate the XCD's L2 per wave and cost
// more than the launch they save: 1 tile 425 -> 412 us, 64 tiles 739 -> 1368 us.)
__global__ __launch_bounds__(256) void ln_small_kernel(const float *__restrict__ Y, const uint4 *__restrict__ resid,
                                                       const float *__restrict__ bias, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, uint4 *__restrict__ act_out) {
    __shared__ float xs[2][4][64];  // [sum | sum of squares][wave][lane]: running values of the halves
    ln4_tile(Y, blockIdx.x, resid, bias, gamma, beta, act_out + (size_t)blockIdx.x * (NFB * 2 * 64), threadIdx.x & 63, threadIdx.x >> 6, xs);
}

__global__ __launch_bounds__(64) MIR_ONE_WAVE void oproj_small_kernel(const uint4 *__restrict__ ctx, const uint4 *__restrict__ wo,
                                                                      float *__restrict__ Y) {
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, tt = blockIdx.y;
    const uint4 *cin = ctx + (size_t)tt * (NFB * 2 * 64) + lane;
    const uint4 *wp = wo + (size_t)nt * (KS_H * 64) + lane;
    uint4 c[KS_H], w[KS_H];  // every load in flight before the first MFMA (hipcc otherwise waits per k-step)
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) { c[ks] = cin[ks * 64]; w[ks] = wp[ks * 64]; }
    __builtin_amdgcn_sched_barrier(0);  // the loads stay up here
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) acc = mfma(w[ks], c[ks], acc);
    store_acc_rows(Y, tt, nt, lane, acc);
}

// one wave: intermediate tile ht of token tile tt from the tile's activation fragments x -> GELU -> h fragments
__device__ __forceinline__ void ffn1_wave(const uint4 (&x)[KS_H], const unsigned char *__restrict__ wffn, const float *__restrict__ b1,
                                          int ht, int tt, uint4 *__restrict__ hbuf, int lane) {
    const int h = lane >> 5;
    const uint4 *wp = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * ht) * (24 * 1024)) + lane;  // W1(ht)
    uint4 w[KS_H];
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) w[ks] = wp[ks * 64];
    float4 bq[4];  // register group g = four consecutive features 8g + 4h ..
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bq[gq] = *reinterpret_cast<const float4 *>(b1 + 32 * ht + 8 * gq + 4 * h);
    __builtin_amdgcn_sched_barrier(0);
    // as in ffn_ln_kernel: the accumulator starts from the bias, GELU by the table behind the parameters (read from L2 here)
    f32x16 acc;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        acc[4 * gq + 0] = bq[gq].x; acc[4 * gq + 1] = bq[gq].y; acc[4 * gq + 2] = bq[gq].z; acc[4 * gq + 3] = bq[gq].w;
    }
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) acc = mfma(w[ks], x[ks], acc);
    const unsigned char *lut = reinterpret_cast<const unsigned char *>(b1 + FFN_PARAM_FLOATS);
    uint32_t off[16], hw[8];
    float xc[16];
    f32x2 e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        off[r] = gelu_lut_addr(acc[r], xc[r], 0u);
        e[r] = *reinterpret_cast<const f32x2 *>(lut + off[r]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) hw[p] = gelu_pack2(acc[2 * p], xc[2 * p], e[2 * p], acc[2 * p + 1], xc[2 * p + 1], e[2 * p + 1]);
    uint4 *ho = hbuf + ((size_t)tt * (2 * NHT) + 2 * ht) * 64 + lane;
    ho[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    ho[64] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
}

__global__ __launch_bounds__(64) MIR_ONE_WAVE void ffn1_small_kernel(const uint4 *__restrict__ act_in,
                                                        const unsigned char *__restrict__ wffn,
                                                        const float *__restrict__ b1, uint4 *__restrict__ hbuf) {
    const int lane = threadIdx.x;
    const int ht = blockIdx.x, tt = blockIdx.y;
    const uint4 *xin = act_in + (size_t)tt * (NFB * 2 * 64) + lane;
    uint4 x[KS_H];
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) x[ks] = xin[ks * 64];
    ffn1_wave(x, wffn, b1, ht, tt, hbuf, lane);
}

// Latency path: the attention block's LayerNorm and FFN1 in one dispatch.  Workgroup (g, tile): its four waves compute
// the tile's LayerNorm together (ln4_tile, into LDS: every one of the 12 workgroups of a tile repeats it - 48 KiB of
// accumulators from L2 each - which costs less than the dispatch it saves), workgroup g = 0 also writes it out (the FFN
// block's residual), then wave w runs intermediate tile 4 g + w from the fragments in LDS.
__global__ __launch_bounds__(256) void ln_ffn1_small_kernel(const float *__restrict__ Y, const uint4 *__restrict__ resid,
                                                            const float *__restrict__ bias, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, uint4 *__restrict__ act_out,
                                                            const unsigned char *__restrict__ wffn, const float *__restrict__ b1,
                                                            uint4 *__restrict__ hbuf) {
    __shared__ float xs[2][4][64];
    __shared__ uint4 xf[NFB * 2 * 64];  // the tile's LayerNorm output, ACT fragments (24 KiB)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tt = blockIdx.y;
    ln4_tile(Y, tt, resid, bias, gamma, beta, xf, lane, w, xs);
    __syncthreads();
    if (blockIdx.x == 0) {
        uint4 *out = act_out + (size_t)tt * (NFB * 2 * 64);
        for (int i = threadIdx.x; i < NFB * 2 * 64; i += 256) out[i] = xf[i];
    }
    uint4 x[KS_H];
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) x[ks] = xf[ks * 64 + lane];
    ffn1_wave(x, wffn, b1, 4 * blockIdx.x + w, tt, hbuf, lane);
}

__global__ __launch_bounds__(64) MIR_ONE_WAVE void ffn2_small_kernel(const uint4 *__restrict__ hbuf,
                                                        const unsigned char *__restrict__ wffn,
                                                        float *__restrict__ Y) {
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, tt = blockIdx.y;
    const uint4 *hin = hbuf + (size_t)tt * (2 * NHT) * 64 + lane;
    // 96 k-steps in 8 chunks of 12, the next chunk's 24 loads in flight while this one's MFMAs run
    constexpr int CH = 6;  // h tiles per chunk (two buffers of 2 x 12 fragments = 192 of the 256 architectural VGPRs)
    uint4 wa[2 * CH], ha[2 * CH], wb[2 * CH], hb2[2 * CH];
    auto load_chunk = [&](int c, uint4 (&w)[2 * CH], uint4 (&hh)[2 * CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ht = c * CH + i;
            // W2(ht) = flat half 2*ht + 1; its fragment for (output tile nt, s2) is piece nt*2 + s2
            const uint4 *wp = reinterpret_cast<const uint4 *>(wffn + (size_t)(2 * ht + 1) * (24 * 1024)) + (nt * 2) * 64 + lane;
            w[2 * i] = wp[0];
            w[2 * i + 1] = wp[64];
            hh[2 * i] = hin[(2 * ht + 0) * 64];
            hh[2 * i + 1] = hin[(2 * ht + 1) * 64];
        }
    };
    f32x16 acc = {0};
    load_chunk(0, wa, ha);
#pragma unroll
    for (int c = 0; c < NHT / CH; c += 2) {
        load_chunk(c + 1, wb, hb2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2 * CH; ++i) acc = mfma(wa[i], ha[i], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NHT / CH) load_chunk(c + 2, wa, ha);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2 * CH; ++i) acc = mfma(wb[i], hb2[i], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    store_acc_rows(Y, tt, nt, lane, acc);
}

// ---------------------------------------------------------------- E5: CLS pooling + L2 normalise
// one wave per sequence: token 0 of the sequence, float32 [384] in natural feature order.
__global__ __launch_bounds__(64) void pool_normalize_kernel(const uint4 *__restrict__ act,
                                                            const int32_t *__restrict__ seq_first_tile, int n_seq,
                                                            int normalize, float *__restrict__ out) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= n_seq) return;
    const uint4 *tile = act + (size_t)seq_first_tile[s] * (NFB * 2 * 64);
    // token 0 of the tile lives in lanes 0 (h=0) and 32 (h=1); lane i < 48 handles fragment block i>>1, half i&1
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float sq = 0.f;
    const int blk = lane >> 1, hh = lane & 1;
    if (lane < 48) {
        frag_to_floats(tile[blk * 64 + 32 * hh], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) sq = fmaf(v[j], v[j], sq);
    }
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float inv = normalize ? 1.0f / fmaxf(sqrtf(sq), 1e-12f) : 1.0f;
    if (lane < 48) {
        const int fb = blk >> 1, s2 = blk & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[(size_t)s * H + 32 * fb + fi(8 * s2 + j, hh)] = v[j] * inv;
    }
}

// debug / test helper: ACT -> float32 [n_tokens][384] natural order; one wave per tile
// (tile_pos: a tile's position in input order - the throughput path orders a pass's tiles its own way, build_batch)
__global__ __launch_bounds__(64) void act_unpack_kernel(const uint4 *__restrict__ act, int n_tiles, const int32_t *__restrict__ tile_pos,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x, h = lane >> 5, t_in = lane & 31;
    if ((int)blockIdx.x >= n_tiles) return;
    const uint4 *tile = act + (size_t)blockIdx.x * (NFB * 2 * 64);
    const int tt = tile_pos[blockIdx.x];
    for (int blk = 0; blk < NFB * 2; ++blk) {
        float v[8];
        frag_to_floats(tile[blk * 64 + lane], v);
        const int fb = blk >> 1, s2 = blk & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[((size_t)tt * 32 + t_in) * H + 32 * fb + fi(8 * s2 + j, h)] = v[j];
    }
}

}  // namespace enc
}  // namespace mir
