// Device code of the bge-small-en (BERT-small) encoder for gfx950.
//
// Replaces the sentence-transformers forward behind
//   aidial_rag/embeddings/embeddings.py:52-108  (bge_embedding_impl, aembed_*)
// 12 layers, hidden 384, 12 heads x 32, FFN 1536, CLS pooling, L2 normalise.
// float16 operands, float32 accumulation (the reference's own CUDA path is
// float16 + SDPA, embeddings.py:43-48).
//
// Design: tokens live on MFMA *columns* (lanes), features on rows (registers).
// Every product is computed transposed, out^T = W^T * x^T, with
// v_mfma_f32_32x32x16_f16: the weight tile is the A operand, 32 tokens' features
// the B operand.  Consequences:
//   * a 32-token tile's activations are 24 B-fragments (96 VGPRs) that stay in
//     registers for a whole kernel; weights are pre-packed on the host into
//     A-fragment order, so a weight k-step is ONE coalesced 1-KiB load (or one
//     LDS-DMA piece) with no transpose, no swizzle, no bank conflict;
//   * the 32x32 accumulator has lane = token, registers = 16 features, which is
//     already the B fragment of the NEXT product up to a fixed permutation of k
//     that is folded into the host-side weight packing ("acc-native" order:
//     element j of lane-half h <-> feature 16*s + 8*(j>>2) + 4*h + (j&3));
//     no LDS round trip between GEMMs, ever;
//   * bias, GELU, softmax statistics, residual and LayerNorm are per-token =
//     per-lane: reductions run over registers plus ONE cross-half shuffle.
//
// Activation layout ACT (float16): [token tile][feature block of 32][s2][64 lanes][8]:
// lane (ti = l&31, h = l>>5), element j of block (fb, s2) = feature
// 32*fb + 16*s2 + 8*(j>>2) + 4*h + (j&3) of token 32*tile + ti.  This is the
// accumulator's own register order, so stores and loads are 16-byte per lane,
// fully coalesced.
#pragma once
// (shared by encoder.hip and encoder_attention.hip: types, fragment helpers, the LayerNorm epilogue)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

namespace mir {
namespace enc {

typedef _Float16 __attribute__((ext_vector_type(8))) f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int H = 384;        // hidden
constexpr int NH = 12;        // heads
constexpr int HD = 32;        // head dim
constexpr int FF = 1536;      // intermediate
constexpr int NFB = H / 32;   // 12 feature blocks
constexpr int KS_H = H / 16;  // 24 k-steps over the hidden dim
constexpr int NHT = FF / 32;  // 48 intermediate tiles
constexpr float LN_EPS = 1e-12f;
// softmax scale 1/sqrt(32) and the log2(e) of exp -> exp2, folded into Q by the QKV kernels (one multiply per Q element
// there instead of one per score in the VALU-bound attention loop)
constexpr float kQScaleLog2e = 0.17677669529663688f * 1.4426950408889634f;
// packed FFN weights: a flat sequence of 24-KiB halves  W1(0) W2(0) W1(1) W2(1) ...  (W1(ht): the 24 k-step fragments
// of intermediate tile ht; W2(ht): the 24 fragments (output tile nt, s2) for k-steps 2*ht + s2); parameters b1 | b2 | gamma | beta
constexpr int FFN_HALF_BYTES = 24 * 1024;
constexpr int FFN_STAGE_BYTES = 2 * FFN_HALF_BYTES;
constexpr int FFN_PARAM_FLOATS = FF + 3 * H;

// feature (row) index inside a 32-row accumulator tile for register r of lane-half h
__device__ __forceinline__ int fi(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
#if defined(ENC_ABL_1616)
    // TIMING-ONLY ablation (results wrong by design): the same operands, flops and matrix-pipe cycles as two
    // v_mfma_f32_16x16x32_f16 - what the chip's clock does with that shape in these kernels (MI355X_MICROARCH.md, DVFS 7)
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c1, 0, 0, 0);
    c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3];
    c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#endif
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    const _Float16 x = (_Float16)a, y = (_Float16)b;
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
// pack2 written as the instruction itself.  Left to hipcc, `(_Float16)fmaf(..)` became v_fma_mixlo_f16
// (ONE rounding, exact -> float16) in ln_small_kernel and fma + v_cvt_pk_f16_f32 (two roundings) in the
// throughput kernels, so the same LayerNorm rounded differently on the two paths in about one value
// per 10^5.  The latency kernel therefore pins the two-rounding form; pinning it in the throughput
// kernels too costs ffn_ln_kernel 34 spilled registers (-15% chunks/s), so those keep the compiler's
// choice and tests/test_gpu_encoder.py::test_batching_is_invariant watches that it stays the same.
__device__ __forceinline__ uint32_t pack2_rn(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The same conversion (v_cvt_pk_f16_f32, round to nearest even) in a form hipcc can SEE: for values that feed an MFMA operand.
// An MFMA that reads a register a vector instruction has just written needs wait states in between; hipcc pads them for
// instructions it knows, never around inline asm (cdna_hip_programming.md 3: "hipcc pads nothing inside asm") - with
// pack2_rn in front of the attention's P x V products the padding was a matter of what the scheduler happened to put in
// between (rounds 1-3: two unrelated instructions; round 4's restructured loop: one, and the MFMA read stale operands).
__device__ __forceinline__ uint32_t pack2_cv(float a, float b) {
    typedef float __attribute__((ext_vector_type(2))) f32x2v;
    typedef _Float16 __attribute__((ext_vector_type(2))) f16x2v;
    const f32x2v v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));
}
// registers 8*s2 .. 8*s2+7 of an accumulator -> one float16 fragment
__device__ __forceinline__ uint4 acc_to_frag(const f32x16 &a, int s2) {
    const int o = 8 * s2;
    return make_uint4(pack2(a[o + 0], a[o + 1]), pack2(a[o + 2], a[o + 3]), pack2(a[o + 4], a[o + 5]),
                      pack2(a[o + 6], a[o + 7]));
}
__device__ __forceinline__ void frag_to_floats(uint4 f, float (&out)[8]) {
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] & 0xffffu));
        out[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] >> 16));
    }
}

__device__ __forceinline__ float half_sum(float x) { return x + __shfl_xor(x, 32, 64); }
__device__ __forceinline__ float half_max(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }

// GELU by table (x * Phi(x), Phi = the standard normal CDF of HF "gelu"): Phi is linear per step of 1/128 on [-5.5, 5.5],
// entry n <-> x_n = -5.5 + n / 128 holds (a_n, b_n) with Phi(x) ~ a_n + b_n x on [x_n - 1/256, x_n + 1/256] (the minimax
// line of the step, built in float64 on the host: gelu_table()); |Phi error| <= 1e-6, so |GELU error| <= 1e-6 |x| - two
// orders below the float16 rounding of the result.  Four full-rate VALU + one 8-byte table read per value instead of
// ~16 VALU + v_rcp + v_exp for the erf formula: the FFN kernel's FFN1 waves are VALU-issue-bound.
constexpr int GELU_LUT_N = 1409;
constexpr int GELU_LUT_FLOATS = 2 * 1410;  // padded to a multiple of 4 floats
constexpr float GELU_LUT_LIM = 5.5f;
// host: the GELU_LUT_FLOATS floats of the table (a_0, b_0, a_1, b_1, ...)
inline void gelu_table(float *out) {
    auto phi = [](double x) { return 0.5 * erfc(-x * 0.70710678118654752440); };
    const double d = 1.0 / 256.0;
    for (int n = 0; n < GELU_LUT_FLOATS / 2; ++n) {
        const double xn = -5.5 + (double)(n < GELU_LUT_N ? n : GELU_LUT_N - 1) / 128.0;
        const double lo = phi(xn - d), hi = phi(xn + d), mid = phi(xn);
        const double b = (hi - lo) / (2.0 * d);
        const double a = 0.5 * (mid + 0.5 * (lo + hi)) - b * xn;  // chord moved half-way to the curve: equal error at the middle and the ends
        out[2 * n] = (float)a;
        out[2 * n + 1] = (float)b;
    }
}
// byte address of x's table entry (base = where the table starts: an LDS byte address or 0 for an offset), and the
// clamped x.  float(2^23 + n) holds n = rne(128 clamp(x) + 704) in its low 24 bits, which v_mad_u32_u24 reads.
__device__ __forceinline__ uint32_t gelu_lut_addr(float x, float &xc, uint32_t base) {
    xc = __builtin_amdgcn_fmed3f(x, -GELU_LUT_LIM, GELU_LUT_LIM);  // NaN -> -5.5 (min3), the product with x restores it
    const float t = fmaf(xc, 128.0f, 8388608.0f + 704.0f);
    uint32_t addr;
    asm("v_mad_u32_u24 %0, %1, 8, %2" : "=v"(addr) : "v"(__float_as_uint(t)), "s"(base));
    return addr;
}
typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;  // a register quad as an inline-asm operand (HIP's uint4 is a struct)
__device__ __forceinline__ f32x2 lds_read_f2(uint32_t byte_addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x2 *>(byte_addr);
}
// float16 pair (lo = gelu(x0), hi = gelu(x1)) from the table entries of x0, x1; the conversion is part of the last
// multiply (v_fma_mix*: one rounding), stated as the instruction so that every kernel rounds alike
__device__ __forceinline__ uint32_t gelu_pack2(float x0, float xc0, f32x2 e0, float x1, float xc1, f32x2 e1) {
    const float p0 = fmaf(e0.y, xc0, e0.x), p1 = fmaf(e1.y, xc1, e1.x);
    uint32_t d;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(x0), "v"(p0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(x1), "v"(p1));
    return d;
}

// LDS-DMA of one 1-KiB piece (16 B per lane; the LDS byte address is wave-uniform), as inline asm so that hipcc's own
// waits do not cover it (see vec_kernels.h: with the builtin every later ds_read is ordered behind ALL pending DMAs)
__device__ __forceinline__ void enc_glds16(const void *gsrc, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// the same with a wave-uniform base in SGPRs and a 32-bit per-lane byte offset: a kernel that issues many pieces per wave keeps
// ONE offset register instead of a 64-bit address per piece
__device__ __forceinline__ void enc_glds16_s(const void *uniform_base, uint32_t lane_byte_off, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_byte_addr)
                 : "memory");
}
// ... non-temporal: for data that is read once (activations), so that it does not push the layer's weights - which every
// workgroup re-reads - out of the L2s (ENC_NT = 0 switches all such hints off, for A/B runs)
#ifndef ENC_NT
#define ENC_NT 1
#endif
__device__ __forceinline__ void enc_glds16_s_nt(const void *uniform_base, uint32_t lane_byte_off, uint32_t lds_byte_addr) {
#if ENC_NT
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_byte_addr)
                 : "memory");
#else
    enc_glds16_s(uniform_base, lane_byte_off, lds_byte_addr);
#endif
}
__device__ __forceinline__ uint4 enc_load_nt(const uint4 *p) {
#if ENC_NT
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void enc_store_nt(uint4 *p, uint4 v) {
#if ENC_NT
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ uint32_t enc_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// Persistent kernels: every workgroup does the same work per group, so all 256 of them reach their HBM-heavy phase (the
// epilogue's residual loads and stores, the next group's activations) at the same moment.  ENC_STAGGER > 0 starts the
// workgroups in four phases, ENC_STAGGER x 64 clocks apart, so that those bursts interleave with the other phases' products.
// Measured per 12288-tile pass (0 / 30 / 60 / 120): oproj_ln_kernel 206.8 / 204.7 / 198.2 / 197.6 us, ffn_ln_kernel 899.0 /
// 897.0 / 893.8 / 893.0 us; the last phase finishes 3 x ENC_STAGGER x 64 clocks (~7 us) after the first.
#ifndef ENC_STAGGER
#define ENC_STAGGER 60
#endif
__device__ __forceinline__ void enc_stagger_start() {
#if ENC_STAGGER > 0
    const int phase = (blockIdx.x >> 3) & 3;  // workgroups go to the 8 XCDs round-robin: phases alternate inside an XCD
    for (int i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(ENC_STAGGER);
#endif
}

// Per-tile bookkeeping: which sequence a 32-token tile belongs to.
struct TileInfo {
    int seq_first_tile;  // first tile of the tile's sequence
    int seq_tiles;       // tiles of that sequence
    int seq_len;         // real tokens of that sequence
    int seq_index;
};

// y[12] (+bias) + residual -> LayerNorm -> ACT store.  `y` rows are features.
// Register r of lane-half h is feature (r&3) + 8*(r>>2) + 4*h of its 32-feature block: the four
// registers of a group g = r>>2 are four CONSECUTIVE features, so bias / gamma / beta are read as
// float4 (a quarter of the loads of the per-register form, which made this epilogue ~3000
// instructions and longer than the output projection's MFMAs).
// The LayerNorm statistics are summed per HALF of the feature blocks (fb 0-5, fb 6-11) and the two halves added, A + B:
// oproj_ln_kernel keeps one half per wave and exchanges exactly these partial sums, and every kernel that calls this
// function must round like it (a sequence's embedding does not depend on which kernels served it).
//
// residual_ln_part: the work of one wave on feature blocks [FB0, FB0 + NB) of a tile: y += bias + residual (in place),
// returns the partial sum; then, given the mean, centres y and returns the partial sum of squares; then, given rstd,
// scales, shifts and stores.  residual_ln_store strings the three steps together for a wave that holds all 12 blocks.
// the residual fragments of blocks [fb0, fb0 + NB): ln_part_sum's loads, for a caller that wants them in flight earlier
template <int NB>
__device__ __forceinline__ void ln_part_load(uint4 (&rr)[NB * 2], int fb0, const uint4 *__restrict__ resid_tile, int lane) {
#pragma unroll
    for (int i = 0; i < NB * 2; ++i) rr[i] = resid_tile[(fb0 * 2 + i) * 64 + lane];
}
// PF: fetch the NEXT block's parameters before working on this one (parameters in global memory: hides their latency, 16
// more live registers here, 32 in ln_part_store); callers whose parameters sit in LDS and whose registers are full pass false.
template <int NB, bool PF = true>
__device__ __forceinline__ float ln_part_sum_rr(f32x16 (&y)[NB], int fb0, const uint4 (&rr)[NB * 2],
                                                const float *__restrict__ bias, int lane, float sum = 0.f) {
    const int h = lane >> 5;
    auto load4 = [&](const float *p, int fb, float4 (&dst)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = *reinterpret_cast<const float4 *>(p + 32 * fb + 8 * g + 4 * h);
    };
    float4 bcur[4], bnext[4];
    if (PF) load4(bias, fb0, bcur);
#pragma unroll
    for (int f = 0; f < NB; ++f) {
        if (PF) { if (f + 1 < NB) load4(bias, fb0 + f + 1, bnext); }
        else load4(bias, fb0 + f, bcur);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float rv[8];
            frag_to_floats(rr[f * 2 + s2], rv);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int g = 2 * s2 + gq;
                const float bb[4] = {bcur[g].x, bcur[g].y, bcur[g].z, bcur[g].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    const float v = y[f][r] + bb[i] + rv[4 * gq + i];
                    y[f][r] = v;
                    sum += v;
                }
            }
        }
        if (PF) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bcur[g] = bnext[g];
        }
    }
    return sum;
}
template <int NB, bool PF = true>
__device__ __forceinline__ float ln_part_sum(f32x16 (&y)[NB], int fb0, const uint4 *__restrict__ resid_tile,
                                             const float *__restrict__ bias, int lane, float sum = 0.f) {
    // Loads first, arithmetic after: written load-next-to-use, hipcc waited for every one of the loads of this
    // epilogue separately (s_waitcnt vmcnt(0) each), which with one wave per SIMD is that many exposed round trips.
    uint4 rr[NB * 2];
    ln_part_load<NB>(rr, fb0, resid_tile, lane);
    return ln_part_sum_rr<NB, PF>(y, fb0, rr, bias, lane, sum);
}
// ln_part_sum without the sum (y += bias + residual only), and the two running sums continued from a given value over
// the blocks in the same element order: for a wave that takes over a half's statistics from another wave part-way
// (ln_small_kernel's four waves per tile).  sum_from(apply(y), 0) == ln_part_sum(y) bit for bit.
template <int NB>
__device__ __forceinline__ void ln_part_apply(f32x16 (&y)[NB], int fb0, const uint4 *__restrict__ resid_tile,
                                              const float *__restrict__ bias, int lane) {
    const int h = lane >> 5;
    uint4 rr[NB * 2];
#pragma unroll
    for (int i = 0; i < NB * 2; ++i) rr[i] = resid_tile[(fb0 * 2 + i) * 64 + lane];
    float4 bb4[NB][4];
#pragma unroll
    for (int f = 0; f < NB; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) bb4[f][g] = *reinterpret_cast<const float4 *>(bias + 32 * (fb0 + f) + 8 * g + 4 * h);
#pragma unroll
    for (int f = 0; f < NB; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float rv[8];
            frag_to_floats(rr[f * 2 + s2], rv);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int g = 2 * s2 + gq;
                const float bb[4] = {bb4[f][g].x, bb4[f][g].y, bb4[f][g].z, bb4[f][g].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) y[f][4 * g + i] = y[f][4 * g + i] + bb[i] + rv[4 * gq + i];
            }
        }
}
template <int NB>
__device__ __forceinline__ float ln_part_sum_from(const f32x16 (&y)[NB], float sum) {
#pragma unroll
    for (int f = 0; f < NB; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += y[f][r];
    return sum;
}
template <int NB>
__device__ __forceinline__ float ln_part_sq_from(const f32x16 (&centred)[NB], float sq) {
#pragma unroll
    for (int f = 0; f < NB; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) sq = fmaf(centred[f][r], centred[f][r], sq);
    return sq;
}
template <int NB>
__device__ __forceinline__ float ln_part_sq(f32x16 (&y)[NB], float mean) {
    float sq = 0.f;
#pragma unroll
    for (int f = 0; f < NB; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dlt = y[f][r] - mean;
            y[f][r] = dlt;
            sq = fmaf(dlt, dlt, sq);
        }
    return sq;
}
template <int NB, bool PIN_CVT, bool PF = true, bool NT = false>
__device__ __forceinline__ void ln_part_store(f32x16 (&y)[NB], int fb0, float rstd, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, uint4 *__restrict__ out_tile, int lane, bool store) {
    const int h = lane >> 5;
    auto load4 = [&](const float *p, int fb, float4 (&dst)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = *reinterpret_cast<const float4 *>(p + 32 * fb + 8 * g + 4 * h);
    };
    float4 gcur[4], ecur[4], gnext[4], enext[4];
    if (PF) {
        load4(gamma, fb0, gcur);
        load4(beta, fb0, ecur);
    }
#pragma unroll
    for (int f = 0; f < NB; ++f) {
        if (!PF) {
            load4(gamma, fb0 + f, gcur);
            load4(beta, fb0 + f, ecur);
        } else if (f + 1 < NB) {
            load4(gamma, fb0 + f + 1, gnext);
            load4(beta, fb0 + f + 1, enext);
        }
        float o[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float gg[4] = {gcur[g].x, gcur[g].y, gcur[g].z, gcur[g].w}, ee[4] = {ecur[g].x, ecur[g].y, ecur[g].z, ecur[g].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) o[4 * g + i] = fmaf(y[f][4 * g + i] * rstd, gg[i], ee[i]);
        }
        if (PF) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { gcur[g] = gnext[g]; ecur[g] = enext[g]; }
        }
        if (store) {
            const int fb = fb0 + f;
            const uint4 w0 =
                PIN_CVT ? make_uint4(pack2_rn(o[0], o[1]), pack2_rn(o[2], o[3]), pack2_rn(o[4], o[5]), pack2_rn(o[6], o[7]))
                        : make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
            const uint4 w1 =
                PIN_CVT ? make_uint4(pack2_rn(o[8], o[9]), pack2_rn(o[10], o[11]), pack2_rn(o[12], o[13]), pack2_rn(o[14], o[15]))
                        : make_uint4(pack2(o[8], o[9]), pack2(o[10], o[11]), pack2(o[12], o[13]), pack2(o[14], o[15]));
            if (NT) {
                enc_store_nt(&out_tile[(fb * 2 + 0) * 64 + lane], w0);
                enc_store_nt(&out_tile[(fb * 2 + 1) * 64 + lane], w1);
            } else {
                out_tile[(fb * 2 + 0) * 64 + lane] = w0;
                out_tile[(fb * 2 + 1) * 64 + lane] = w1;
            }
        }
    }
}

// y[12] (+bias) + residual -> LayerNorm -> ACT store, for a wave that holds the whole tile.  `y` rows are features.
// Register r of lane-half h is feature (r&3) + 8*(r>>2) + 4*h of its 32-feature block: the four registers of a group
// g = r>>2 are four CONSECUTIVE features, so bias / gamma / beta are read as float4.
// QUARTERS: load the residual three blocks at a time instead of six (a wave that also holds other state: the FFN kernel);
// the sums run over the blocks in the same order either way, so the result is the same bit for bit.
template <bool PIN_CVT = false, bool QUARTERS = false, bool PF = true>
__device__ __forceinline__ void residual_ln_store(f32x16 (&y)[NFB], const uint4 *__restrict__ resid_tile,
                                                  const float *__restrict__ bias, const float *__restrict__ gamma,
                                                  const float *__restrict__ beta, uint4 *__restrict__ out_tile,
                                                  int lane, bool store) {
    constexpr int HB = NFB / 2;
    f32x16(&ya)[HB] = *reinterpret_cast<f32x16(*)[HB]>(&y[0]);
    f32x16(&yb)[HB] = *reinterpret_cast<f32x16(*)[HB]>(&y[HB]);
    float sa, sb;
    if (QUARTERS) {
        constexpr int QB = HB / 2;
        f32x16(&y0)[QB] = *reinterpret_cast<f32x16(*)[QB]>(&y[0]);
        f32x16(&y1)[QB] = *reinterpret_cast<f32x16(*)[QB]>(&y[QB]);
        f32x16(&y2)[QB] = *reinterpret_cast<f32x16(*)[QB]>(&y[2 * QB]);
        f32x16(&y3)[QB] = *reinterpret_cast<f32x16(*)[QB]>(&y[3 * QB]);
        sa = ln_part_sum<QB, PF>(y0, 0, resid_tile, bias, lane);
        sa = ln_part_sum<QB, PF>(y1, QB, resid_tile, bias, lane, sa);
        sb = ln_part_sum<QB, PF>(y2, 2 * QB, resid_tile, bias, lane);
        sb = ln_part_sum<QB, PF>(y3, 3 * QB, resid_tile, bias, lane, sb);
    } else {
        sa = ln_part_sum<HB, PF>(ya, 0, resid_tile, bias, lane);
        sb = ln_part_sum<HB, PF>(yb, HB, resid_tile, bias, lane);
    }
    const float mean = half_sum(sa + sb) * (1.0f / H);
    const float qa = ln_part_sq<HB>(ya, mean);
    const float qb = ln_part_sq<HB>(yb, mean);
    const float rstd = rsqrtf(half_sum(qa + qb) * (1.0f / H) + LN_EPS);
    ln_part_store<HB, PIN_CVT, PF>(ya, 0, rstd, gamma, beta, out_tile, lane, store);
    ln_part_store<HB, PIN_CVT, PF>(yb, HB, rstd, gamma, beta, out_tile, lane, store);
}


// The latency path's LayerNorm of one token tile by a workgroup's first four waves (ln_small_kernel and the fused kernels
// that repeat it: encoder_kernels.h, encoder_attention.hip).
// (the body: four waves, w = threadIdx.x >> 6; `out_tile` may be global memory or LDS; xs = [2][4][64] floats of LDS)
__device__ __forceinline__ void ln4_tile(const float *__restrict__ Y, int tt, const uint4 *__restrict__ resid,
                                         const float *__restrict__ bias, const float *__restrict__ gamma,
                                         const float *__restrict__ beta, uint4 *out_tile, int lane, int w, float (*xs)[4][64]) {
    constexpr int QB = NFB / 4;
    const bool second = w & 1;
    f32x16 y[QB];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const float4 *yi = reinterpret_cast<const float4 *>(Y + (((size_t)tt * NFB + QB * w + i) * 16) * 64) + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = yi[c * 64];
            y[i][4 * c + 0] = v.x; y[i][4 * c + 1] = v.y; y[i][4 * c + 2] = v.z; y[i][4 * c + 3] = v.w;
        }
    }
    ln_part_apply<QB>(y, QB * w, resid + (size_t)tt * (NFB * 2 * 64), bias, lane);
    if (!second) xs[0][w][lane] = ln_part_sum_from<QB>(y, 0.f);
    __syncthreads();
    if (second) xs[0][w][lane] = ln_part_sum_from<QB>(y, xs[0][w - 1][lane]);
    __syncthreads();
    const float mean = half_sum(xs[0][1][lane] + xs[0][3][lane]) * (1.0f / H);  // half A + half B
#pragma unroll
    for (int f = 0; f < QB; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[f][r] = y[f][r] - mean;
    if (!second) xs[1][w][lane] = ln_part_sq_from<QB>(y, 0.f);
    __syncthreads();
    if (second) xs[1][w][lane] = ln_part_sq_from<QB>(y, xs[1][w - 1][lane]);
    __syncthreads();
    const float rstd = rsqrtf(half_sum(xs[1][1][lane] + xs[1][3][lane]) * (1.0f / H) + LN_EPS);
    ln_part_store<QB, true>(y, QB * w, rstd, gamma, beta, out_tile, lane, true);
}


}  // namespace enc
}  // namespace mir
