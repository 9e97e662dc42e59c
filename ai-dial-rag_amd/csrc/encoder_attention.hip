// Attention kernel of the bge-small-en encoder - see encoder_common.h for the layouts and why this
// kernel has a translation unit (and one compiler option) of its own.
#include "common.h"
#include "encoder_common.h"

namespace mir {
namespace enc {

// The running state of one (head, query tile) and one key tile's worth of attention: S^T = K Q^T - ref, P = exp2,
// O^T += V^T P^T, lsum += 1 P^T (the lazy softmax reference is described at attention_kernel).  Shared by the throughput /
// general kernel and the single-tile fused kernel of the latency path, so both round alike.
typedef __attribute__((ext_vector_type(4))) float f32x4;
struct AttnState {
    f32x16 o = {0}, nref = {0};
    f32x4 lsum = {0};  // every register holds the lane's query's running sum
};
// Row sums of P on the matrix pipe: a 16x16x32 product of a 0 / 1 SELECTOR with the P^T fragment as it stands.  As that
// instruction's B operand the fragment's lane l is "column l & 15, k-slice l >> 4"; it really holds query l & 31, keys
// 8 (l >> 5) ..: lanes n, n + 32 are query n's two key halves, lanes n + 16, n + 48 query n + 16's.  The selector's row i takes
// the k-slices g with g & 1 == (i >> 2) & 1, and lane l reads back rows 4 (l >> 4) ..: its OWN query's sum over the fragment's
// 16 keys, four times.  (Round 3 multiplied ones by P^T on the 32x32x16 shape: 16 identical registers per query tile and
// twice the matrix cycles; two query tiles per wave need the registers.)
__device__ __forceinline__ uint4 attn_sum_selector(int lane) {
    const uint32_t w = (((lane >> 4) & 1) == ((lane >> 2) & 1)) ? 0x3C003C00u : 0u;
    return make_uint4(w, w, w, w);
}
__device__ __forceinline__ f32x4 mfma_sum(uint4 sel, uint4 p, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sel), __builtin_bit_cast(f16x8, p), c, 0, 0, 0);
}
constexpr float kAttnSlack = 6.0f;
// ATTN_ABL: TIMING-ONLY ablations (results wrong by design; tools/run_enc_variants.sh): bit 0 no exponentials, bit 1 no reference
// check (no max tree, no vote), bit 2 no P x V / row-sum MFMAs, bit 3 no score MFMAs, bit 4 every key tile is the sequence's first
// (K / V from the caches), bit 5 no context store
#ifndef ATTN_ABL
#define ATTN_ABL 0
#endif
// The pieces of one (query tile, key tile) step; attn_step_t strings them together for one query tile, attn_step2_t for the
// two query tiles of a wave (same operations on the same values in the same order per query tile: both round alike).
// FIRST / LAST are compile-time: the kernels peel a sequence's first and last key tile out of their loops, so the steady state
// carries no masking code and no branch but the rare reference move.
//
// Padding (only a sequence's LAST key tile can hold it) is masked through the score MFMA's C operand: -inf in the rows of
// keys past the sequence's end instead of -ref (x - inf = -inf: the same value the old select wrote after the MFMAs).
template <bool LAST>
__device__ __forceinline__ f32x16 attn_scores(const AttnState &st, const uint4 &q0, const uint4 &q1, const uint4 &kc0, const uint4 &kc1,
                                              int key0, int seq_len, int h) {
#if defined(ATTN_MASK_AFTER)
    f32x16 s = mfma(kc0, q0, st.nref);
    s = mfma(kc1, q1, s);
    if (LAST) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (key0 + fi(r, h) < seq_len) ? s[r] : -__builtin_inff();
    }
    return s;
#else
    f32x16 c = st.nref;
    if (LAST) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = (key0 + fi(r, h) < seq_len) ? c[r] : -__builtin_inff();
    }
#if ATTN_ABL & 8
    f32x16 s = c;
    s[0] += __uint_as_float(kc0.x & q0.x & 0xffu) + __uint_as_float(kc1.x & q1.x & 0xffu);
    return s;
#else
    f32x16 s = mfma(kc0, q0, c);
    return mfma(kc1, q1, s);
#endif
#endif
}
__device__ __forceinline__ float attn_lane_max(const f32x16 &s) {  // this lane's half of the keys
    // (fmaxf: each MFMA output comes with a canonicalising v_max x, x.  An inline-asm v_max3_f32 does not - and is WRONG here:
    // hipcc's hazard recogniser does not look into inline asm, so the wait states between an MFMA and a VALU read of its
    // result are not inserted; measured: garbage scores.  The VALU count is not what bounds this kernel anyway.)
    const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]), a2 = fmaxf(fmaxf(s[6], s[7]), s[8]);
    const float a3 = fmaxf(fmaxf(s[9], s[10]), s[11]), a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
    return fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(a4, s[15]));
}
// move the reference by delta (0 leaves every value as it is, bit for bit: x - 0, x * exp2(-0))
__device__ __forceinline__ void attn_rescale(AttnState &st, f32x16 &s, float delta) {
    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s[r] -= delta;
        st.nref[r] -= delta;
        st.o[r] *= alpha;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) st.lsum[r] *= alpha;
}
__device__ __forceinline__ void attn_pv(AttnState &st, f32x16 &s, const uint4 &vc0, const uint4 &vc1) {
    const uint4 sel = attn_sum_selector((int)(threadIdx.x & 63));
    // v_exp_f32 directly: exp2f() wraps it in a compare / select / ldexp to keep results below 2^-126 exact, ~4 extra
    // instructions per value; a softmax term that small is zero next to the row's largest term either way
#if !(ATTN_ABL & 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
#endif
    // P to float16 with the packed convert (two values per instruction; the scalar casts cost three)
    // (pack2_cv, not the inline-asm pack2_rn: these registers are MFMA operands - see pack2_cv)
    const uint4 p0 = make_uint4(pack2_cv(s[0], s[1]), pack2_cv(s[2], s[3]), pack2_cv(s[4], s[5]), pack2_cv(s[6], s[7]));
    const uint4 p1 = make_uint4(pack2_cv(s[8], s[9]), pack2_cv(s[10], s[11]), pack2_cv(s[12], s[13]), pack2_cv(s[14], s[15]));
#if ATTN_ABL & 4
    st.o[0] += __uint_as_float((p0.x ^ p1.y ^ vc0.x ^ vc1.x ^ p0.z ^ p1.w ^ p0.y ^ p1.x ^ p0.w ^ p1.z ^ sel.x) & 0xffu);
    st.lsum[0] += 1.0f;
#else
    st.o = mfma(vc0, p0, st.o);
    st.lsum = mfma_sum(sel, p0, st.lsum);
    st.o = mfma(vc1, p1, st.o);
    st.lsum = mfma_sum(sel, p1, st.lsum);
#endif
}
template <bool FIRST, bool LAST>
__device__ __forceinline__ void attn_step_t(AttnState &st, const uint4 &q0, const uint4 &q1, const uint4 &kc0, const uint4 &kc1,
                                            const uint4 &vc0, const uint4 &vc1, int key0, int seq_len, int h) {
    f32x16 s = attn_scores<LAST>(st, q0, q1, kc0, kc1, key0, seq_len, h);
    const float mx = attn_lane_max(s);
    if (FIRST || __any(mx > kAttnSlack)) {  // wave-uniform; rare after the first tile
        const float mq = half_max(mx);
        attn_rescale(st, s, FIRST ? mq : fmaxf(mq, 0.f));  // the reference only rises after the first tile
    }
    attn_pv(st, s, vc0, vc1);
}
// the single-tile kernels' step (first and last at once)
__device__ __forceinline__ void attn_step(AttnState &st, const uint4 &q0, const uint4 &q1, const uint4 &kc0, const uint4 &kc1,
                                          const uint4 &vc0, const uint4 &vc1, bool first, bool last, int key0, int seq_len, int h) {
    if (first && last) attn_step_t<true, true>(st, q0, q1, kc0, kc1, vc0, vc1, key0, seq_len, h);
    else if (first) attn_step_t<true, false>(st, q0, q1, kc0, kc1, vc0, vc1, key0, seq_len, h);
    else if (last) attn_step_t<false, true>(st, q0, q1, kc0, kc1, vc0, vc1, key0, seq_len, h);
    else attn_step_t<false, false>(st, q0, q1, kc0, kc1, vc0, vc1, key0, seq_len, h);
}
// Two query tiles of one (sequence, head) against the same key tile: the K / V fragments are fetched once for both, and the
// two tiles' chains are independent, so one's exponentials issue beside the other's MFMAs.  The rare branch is shared: a tile
// whose own condition is false gets delta = 0, which changes nothing (attn_rescale), so each tile's values are exactly
// attn_step_t's.
template <bool FIRST, bool LAST>
__device__ __forceinline__ void attn_step2_t(AttnState &sa, AttnState &sb, const uint4 (&qa)[2], const uint4 (&qb)[2], const uint4 &kc0,
                                             const uint4 &kc1, const uint4 &vc0, const uint4 &vc1, int key0, int seq_len, int h) {
    f32x16 s0 = attn_scores<LAST>(sa, qa[0], qa[1], kc0, kc1, key0, seq_len, h);
    f32x16 s1 = attn_scores<LAST>(sb, qb[0], qb[1], kc0, kc1, key0, seq_len, h);
    const float mx0 = attn_lane_max(s0), mx1 = attn_lane_max(s1);
    const bool any0 = FIRST || __any(mx0 > kAttnSlack), any1 = FIRST || __any(mx1 > kAttnSlack);
    if (any0 || any1) {
        const float mq0 = half_max(mx0), mq1 = half_max(mx1);
        attn_rescale(sa, s0, FIRST ? mq0 : any0 ? fmaxf(mq0, 0.f) : 0.f);
        attn_rescale(sb, s1, FIRST ? mq1 : any1 ? fmaxf(mq1, 0.f) : 0.f);
    }
    attn_pv(sa, s0, vc0, vc1);
    attn_pv(sb, s1, vc0, vc1);
}
// attn_step2_t with run-time flags (ONE body; the A/B partner of the peeled loop: -DATTN_PEEL=0)
__device__ __forceinline__ void attn_step2_rt(AttnState &sa, AttnState &sb, const uint4 (&qa)[2], const uint4 (&qb)[2], const uint4 &kc0,
                                              const uint4 &kc1, const uint4 &vc0, const uint4 &vc1, bool first, bool last, int key0,
                                              int seq_len, int h) {
    f32x16 s0 = attn_scores<false>(sa, qa[0], qa[1], kc0, kc1, key0, seq_len, h);
    f32x16 s1 = attn_scores<false>(sb, qb[0], qb[1], kc0, kc1, key0, seq_len, h);
    if (last) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = key0 + fi(r, h) < seq_len;
            s0[r] = ok ? s0[r] : -__builtin_inff();
            s1[r] = ok ? s1[r] : -__builtin_inff();
        }
    }
    const float mx0 = attn_lane_max(s0), mx1 = attn_lane_max(s1);
    const bool any0 = first || __any(mx0 > kAttnSlack), any1 = first || __any(mx1 > kAttnSlack);
    if (any0 || any1) {
        const float mq0 = half_max(mx0), mq1 = half_max(mx1);
        attn_rescale(sa, s0, first ? mq0 : any0 ? fmaxf(mq0, 0.f) : 0.f);
        attn_rescale(sb, s1, first ? mq1 : any1 ? fmaxf(mq1, 0.f) : 0.f);
    }
    attn_pv(sa, s0, vc0, vc1);
    attn_pv(sb, s1, vc0, vc1);
}
// One-body step with run-time flags for ONE query tile (attn_step's four instantiations cost attention_kernel a wave per SIMD: 147 VGPRs)
__device__ __forceinline__ void attn_step1_rt(AttnState &st, const uint4 (&q)[2], const uint4 &kc0, const uint4 &kc1, const uint4 &vc0,
                                              const uint4 &vc1, bool first, bool last, int key0, int seq_len, int h) {
    f32x16 s = attn_scores<false>(st, q[0], q[1], kc0, kc1, key0, seq_len, h);
    if (last) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (key0 + fi(r, h) < seq_len) ? s[r] : -__builtin_inff();
    }
#if ATTN_ABL & 2
    if (first) attn_rescale(st, s, half_max(s[0]));
#else
    const float mx = attn_lane_max(s);
    if (first || __any(mx > kAttnSlack)) {
        const float mq = half_max(mx);
        attn_rescale(st, s, first ? mq : fmaxf(mq, 0.f));
    }
#endif
    attn_pv(st, s, vc0, vc1);
}

// normalise and write the context fragments of the (head, query tile)
__device__ __forceinline__ void attn_store(AttnState &st, uint4 *__restrict__ out) {
    const float inv = 1.0f / st.lsum[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) st.o[r] *= inv;
#if ATTN_ABL & 32
    if (st.o[0] == 1234.5f && st.o[9] == 77.25f)
#endif
    {
    enc_store_nt(out, acc_to_frag(st.o, 0));  // the context is read once, by the output projection (measured: within noise
    enc_store_nt(out + 64, acc_to_frag(st.o, 1));  // for this kernel, 361 vs 364 us; the projection after it 197 vs 202 us)
    }
}

// One wave per (head, query tile): light on registers, so several waves share a SIMD and
// the softmax's VALU work overlaps other waves' MFMAs (at hd = 32 a 32x32 score tile is 4
// MFMAs against ~100 VALU instructions: a one-wave-per-SIMD kernel is VALU-bound 4:1).
// S^T = K Q^T keeps keys on rows, so the softmax statistics of a query are lane-local plus
// one cross-half shuffle; P^T (converted in registers) is the B operand of O^T += V^T P^T.
// Output: the context in ACT layout (feature block = head), read by the projection kernel.
__global__ __launch_bounds__(256) void attention_kernel(const uint4 *__restrict__ qf, const uint4 *__restrict__ kf,
                                                        const uint4 *__restrict__ vf,
                                                        const TileInfo *__restrict__ ti, int n_tiles,
                                                        uint4 *__restrict__ ctx) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    // Workgroups go to the 8 XCDs round-robin (blockIdx & 7), each with its own L2.  Every query tile of a sequence reads
    // the sequence's K/V of its head, so the tiles of one (head, sequence) must meet in ONE L2: XCD x takes the x-th
    // eighth of the (head, tile) list instead of every eighth entry.  (Counters, tools/run_enc_pmc.sh: with the plain
    // mapping 73 % of the kernel's L2 requests missed and it pulled 0.53 GB per 2888-tile pass through the fabric -
    // ~5 TB/s: the kernel's whole time - where Q + K + V are 0.21 GB.)
    const int n_wg = gridDim.x, per_xcd = (n_wg + 7) >> 3;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int w = wg * 4 + (threadIdx.x >> 6);
    if ((blockIdx.x >> 3) >= per_xcd || w >= n_tiles * NH) return;
    const int head = w / n_tiles, tt = w - head * n_tiles;   // neighbours share a sequence's K/V of one head
    const TileInfo info = ti[tt];
    // Q arrives pre-multiplied by log2(e) / sqrt(32) (qkv kernels): the scores are exp2 arguments as they leave the MFMA
    const uint4 *qp = qf + ((size_t)(tt * NH + head) * 2) * 64 + lane;
    const uint4 q0 = enc_load_nt(qp), q1 = enc_load_nt(qp + 64);  // Q is read once; K / V are shared by the sequence's query tiles
    const int n_kt = __builtin_amdgcn_readfirstlane(info.seq_tiles);  // a tile's bookkeeping is wave-uniform: scalar loop control
    const int seq_len = __builtin_amdgcn_readfirstlane(info.seq_len);
    const size_t kv0 = ((size_t)(__builtin_amdgcn_readfirstlane(info.seq_first_tile) * NH + head) * 2) * 64 + lane;

    // Online softmax with a LAZY reference: a query's scores are taken relative to a reference value ref (exp2 arguments
    // s - ref), which moves only when some score exceeds it by more than kSlack (or on the first key tile, to that tile's
    // maximum).  The usual running maximum moves for SOME of a wave's 32 queries on nearly every tile, and every move costs
    // the wave the rescale of its accumulators; here the steady state of a tile is the score MFMAs, a max tree, one
    // vote, 16 v_exp, 8 packed converts and the output MFMAs:
    //   * -ref sits in a 16-register tile that is the C operand of the score MFMA: no per-score subtraction;
    //   * the row sums come from the matrix pipe (ones x P^T into a second accumulator): no adds, no cross-half shuffle;
    //   * P stays <= 2^kSlack: float16 holds it with the same relative precision as values <= 1.
    // (~110 -> ~50 VALU instructions per key tile, tools/run_enc_pmc.sh.)
    AttnState st;
    const uint4 qq[2] = {q0, q1};
    auto step = [&](int kt, const uint4 &kc0, const uint4 &kc1, const uint4 &vc0, const uint4 &vc1) {
        attn_step1_rt(st, qq, kc0, kc1, vc0, vc1, kt == 0, kt == n_kt - 1, 32 * kt, seq_len, h);
    };
    // Two key tiles per trip, each in its own registers: a tile's K/V are requested one step ahead.  vmcnt counts in
    // order, so the wait for the older buffer leaves the younger one's loads in flight.  (The requests are unconditional -
    // past the end they re-read the last tile: behind a branch that may or may not have issued loads, hipcc waits for
    // vmcnt(0).)
    auto load = [&](int kt, uint4 &k0, uint4 &k1, uint4 &v0, uint4 &v1) {
        const size_t kb = kv0 + (size_t)((ATTN_ABL & 16) ? 0 : kt) * (NH * 2 * 64);
        k0 = kf[kb]; k1 = kf[kb + 64]; v0 = vf[kb]; v1 = vf[kb + 64];
    };
    uint4 ka0, ka1, va0, va1, kb0, kb1, vb0, vb1;
    load(0, ka0, ka1, va0, va1);
    for (int kt = 0; kt < n_kt; kt += 2) {
        load(min(kt + 1, n_kt - 1), kb0, kb1, vb0, vb1);
        step(kt, ka0, ka1, va0, va1);
        if (kt + 1 < n_kt) {
            load(min(kt + 2, n_kt - 1), ka0, ka1, va0, va1);
            step(kt + 1, kb0, kb1, vb0, vb1);
        }
    }
    attn_store(st, ctx + (size_t)tt * (NFB * 2 * 64) + (size_t)(head * 2) * 64 + lane);
}


// The throughput form (round 4): one wave per (head, UNIT), a unit = two consecutive query tiles of one sequence (its last unit
// one tile when the sequence has an odd number).  `units[u]` = {first tile | (tiles - 1) << 24, key tiles, sequence length, the sequence's first tile}, built on the host.
// By the counters the one-tile kernel above drew 24 B/clk per CU from L2 - each (query tile, key tile) step re-reads 4 KiB of
// K / V - which is what a CU can draw (26-29 B/clk); two query tiles per fetch halve that.
template <int NQ>
__device__ __forceinline__ void attn_unit(const uint4 *__restrict__ qf, const uint4 *__restrict__ kf, const uint4 *__restrict__ vf,
                                          int tt, int n_kt, int seq_len, int seq_first_tile, int head, int lane, uint4 *__restrict__ ctx) {
    const int h = lane >> 5;
    uint4 qa[2], qb[2];
    {
        const uint4 *qp = qf + ((size_t)(tt * NH + head) * 2) * 64 + lane;
        qa[0] = enc_load_nt(qp); qa[1] = enc_load_nt(qp + 64);
        if (NQ == 2) { qb[0] = enc_load_nt(qp + NH * 2 * 64); qb[1] = enc_load_nt(qp + NH * 2 * 64 + 64); }
    }
    const size_t kv0 = ((size_t)(seq_first_tile * NH + head) * 2) * 64 + lane;
    AttnState sa, sb;
    struct KV { uint4 k0, k1, v0, v1; };
    auto load = [&](int kt, KV &d) {
        const size_t kb = kv0 + (size_t)kt * (NH * 2 * 64);
        d.k0 = kf[kb]; d.k1 = kf[kb + 64]; d.v0 = vf[kb]; d.v1 = vf[kb + 64];
    };
#define ATTN_STEP(FIRST, LAST, kt, b)                                                                              \
    do {                                                                                                           \
        if (NQ == 2) attn_step2_t<FIRST, LAST>(sa, sb, qa, qb, b.k0, b.k1, b.v0, b.v1, 32 * (kt), seq_len, h);     \
        else attn_step_t<FIRST, LAST>(sa, qa[0], qa[1], b.k0, b.k1, b.v0, b.v1, 32 * (kt), seq_len, h);           \
    } while (0)
    // The first and the last key tile are peeled (attn_step_t); in between two key tiles per trip, each in its own registers,
    // a tile's K / V requested one step ahead: vmcnt counts in order, so the wait for the older buffer leaves the younger one's
    // loads in flight.
#ifndef ATTN_PEEL
#define ATTN_PEEL 1
#endif
    KV A, B;
    load(0, A);
#if !ATTN_PEEL
    for (int kt = 0; kt < n_kt; kt += 2) {
        load(min(kt + 1, n_kt - 1), B);
        if (NQ == 2) attn_step2_rt(sa, sb, qa, qb, A.k0, A.k1, A.v0, A.v1, kt == 0, kt == n_kt - 1, 32 * kt, seq_len, h);
        else attn_step(sa, qa[0], qa[1], A.k0, A.k1, A.v0, A.v1, kt == 0, kt == n_kt - 1, 32 * kt, seq_len, h);
        if (kt + 1 < n_kt) {
            load(min(kt + 2, n_kt - 1), A);
            if (NQ == 2) attn_step2_rt(sa, sb, qa, qb, B.k0, B.k1, B.v0, B.v1, false, kt + 1 == n_kt - 1, 32 * (kt + 1), seq_len, h);
            else attn_step(sa, qa[0], qa[1], B.k0, B.k1, B.v0, B.v1, false, kt + 1 == n_kt - 1, 32 * (kt + 1), seq_len, h);
        }
    }
#else
    if (n_kt == 1) {
        ATTN_STEP(true, true, 0, A);
    } else {
        load(1, B);
        ATTN_STEP(true, false, 0, A);
        int kt = 1;  // B holds tile kt
        while (kt + 2 < n_kt) {
            load(kt + 1, A);
            ATTN_STEP(false, false, kt, B);
            load(kt + 2, B);
            ATTN_STEP(false, false, kt + 1, A);
            kt += 2;
        }
        if (kt + 1 < n_kt) {
            load(kt + 1, A);
            ATTN_STEP(false, false, kt, B);
            ATTN_STEP(false, true, kt + 1, A);
        } else {
            ATTN_STEP(false, true, kt, B);
        }
    }
#endif
#undef ATTN_STEP
    uint4 *out = ctx + (size_t)tt * (NFB * 2 * 64) + (size_t)(head * 2) * 64 + lane;
    attn_store(sa, out);
    if (NQ == 2) attn_store(sb, out + NFB * 2 * 64);
}
__global__ __launch_bounds__(256) void attention2_kernel(const uint4 *__restrict__ qf, const uint4 *__restrict__ kf,
                                                         const uint4 *__restrict__ vf, const int4 *__restrict__ units, int n_units,
                                                         uint4 *__restrict__ ctx) {
    const int lane = threadIdx.x & 63;
    // XCD x takes the x-th eighth of the (head, unit) list, as attention_kernel: the units of one (head, sequence) meet in one L2
    const int n_wg = gridDim.x, per_xcd = (n_wg + 7) >> 3;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int w = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));  // wave-uniform: scalar arithmetic, scalar descriptor load
    if ((blockIdx.x >> 3) >= per_xcd || w >= n_units * NH) return;
    const int head = w / n_units, u = w - head * n_units;
    // the unit's descriptor in ONE wave-uniform (scalar) load: {first tile | (tiles - 1) << 24, key tiles, sequence length,
    // the sequence's first tile} - a wave lives for a handful of key tiles, so every dependent hop in front of its first
    // MFMA (descriptor -> tile info -> Q / K addresses were three) is a large share of its life
    typedef int __attribute__((ext_vector_type(4))) i32x4;
    const i32x4 d = *reinterpret_cast<const __attribute__((address_space(4))) i32x4 *>(reinterpret_cast<uintptr_t>(units + u));
    const int tt = d.x & 0xffffff;
    if (d.x >> 24) attn_unit<2>(qf, kf, vf, tt, d.y, d.z, d.w, head, lane, ctx);
    else attn_unit<1>(qf, kf, vf, tt, d.y, d.z, d.w, head, lane, ctx);
}


// FUSED QKV projection + attention (round 4, ENC_FQA): Q, K and V never leave the CU.
// Per layer and 12 288-tile pass qkv_kernel writes 0.9 GB of Q / K / V and attention_kernel reads them back - K / V once per
// query tile of the sequence - and by counters and ablations both kernels sit on the CU's vector-memory path (~28 B/clk per
// CU: attention 367 us with real K / V loads and no arithmetic at all 337).  Here a workgroup takes a BIN of whole
// sequences, at most FQA_TILES token tiles, one WAVE per tile:
//   * the wave keeps its tile's activations resident (24 B fragments, 96 VGPRs) through all 12 heads;
//   * the layer's 36 weight blocks (24 KiB: one head's K, V or Q projection) stream through a 3-slot LDS ring (LDS-DMA, one
//     counted wait + one raw barrier per block), shared by the bin's waves, in head order K_h, V_h, Q_h;
//   * per head every wave projects its tile's K and V (the same chains and epilogue arithmetic as qkv_kernel / qkv_small_kernel:
//     a sequence's embedding does not depend on which kernels served it), leaves the fragments in LDS, projects Q into
//     registers, and after one barrier attends over its sequence's key tiles FROM LDS (attn_step1_rt, the kernels' shared step);
//   * the context goes out as attention_kernel writes it.
// What crosses the vector-memory path per tile and layer: x once (24 KiB), the weights once per bin (864 KiB / 8 tiles), the
// context (2 KiB per head) - against x + 72 KiB of Q / K / V out and (2 + 4 n_kt) KiB per head in.
// Sequences of more than FQA_TILES tiles stay on qkv_kernel + attention_kernel (their activations do not fit the registers
// of one workgroup): the host puts them FIRST in a pass, the binned ones after.
constexpr int FQA_TILES = 8;                         // waves per workgroup = token tiles per bin
constexpr int FQA_WSLOTS = 3;
#ifndef FQA_PF
#define FQA_PF 6
#endif
// FQA_ABL: TIMING-ONLY ablations of the fused kernel (results wrong by design): bit 0 one key tile per head instead of the
// sequence's n_kt, bit 1 a third of the projections' k-steps
#ifndef FQA_ABL
#define FQA_ABL 0
#endif
constexpr int FQA_BLOCK_BYTES = KS_H * 1024;         // one (projection, head) weight block
constexpr int FQA_PPW = KS_H / FQA_TILES;            // 1-KiB pieces of a block per wave
constexpr int FQA_KV_OFF = FQA_WSLOTS * FQA_BLOCK_BYTES;
constexpr int FQA_BIAS_OFF = FQA_KV_OFF + FQA_TILES * 4096;
constexpr int FQA_LDS_BYTES = FQA_BIAS_OFF + 3 * H * 4;
static_assert(KS_H % FQA_TILES == 0, "a weight block's pieces divide among the waves");

__global__ __launch_bounds__(64 * FQA_TILES, 1) void fused_qkv_attention_kernel(const uint4 *__restrict__ act, const uint4 *__restrict__ wqkv,
                                                                                 const float *__restrict__ bqkv, const int4 *__restrict__ bins,
                                                                                 uint4 *__restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *bias = reinterpret_cast<float *>(smem + FQA_BIAS_OFF);
    uint4 *kvs = reinterpret_cast<uint4 *>(smem + FQA_KV_OFF);  // [slot][K0 K1 V0 V1][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave's slot of the bin: {tile or -1, first slot of its sequence | key tiles << 8, sequence length, -}
    typedef int __attribute__((ext_vector_type(4))) i32x4;
    const i32x4 sl = *reinterpret_cast<const __attribute__((address_space(4))) i32x4 *>(reinterpret_cast<uintptr_t>(bins + (size_t)blockIdx.x * FQA_TILES + wave));
    const bool live = sl.x >= 0;
    const i32x4 s0 = *reinterpret_cast<const __attribute__((address_space(4))) i32x4 *>(reinterpret_cast<uintptr_t>(bins + (size_t)blockIdx.x * FQA_TILES));
    const int tt = live ? sl.x : s0.x;  // an empty slot shadows the bin's first tile: it moves its share of the weights, stores nothing
    const int kv0 = sl.y & 0xff, n_kt = sl.y >> 8, seq_len = sl.z;
    for (int i = tid; i < 3 * H; i += 64 * FQA_TILES) bias[i] = bqkv[i];
    uint4 x[KS_H];
    {
        const uint4 *xin = act + (size_t)tt * (NFB * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) x[ks] = enc_load_nt(xin + ks * 64);  // read once here (the output projection's LayerNorm re-reads it as the residual much later)
    }
    // ordinary loads are complete before the first DMA: the counted waits below count DMAs and stores only
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) asm volatile("" : "+v"(x[ks].x), "+v"(x[ks].y), "+v"(x[ks].z), "+v"(x[ks].w));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the biases are in LDS

    // weight blocks in the order they are used: head hd: K (tile 12 + hd), V (24 + hd), Q (hd)
    auto block_tile = [](int wb) { const int hd = wb / 3, j = wb - 3 * hd; return (j == 0 ? NH : j == 1 ? 2 * NH : 0) + hd; };
    auto issue = [&](int wb) {
        const uint4 *src = wqkv + (size_t)block_tile(wb) * (KS_H * 64) + (size_t)(wave * FQA_PPW) * 64;  // wave-uniform
        const uint32_t dst = __builtin_amdgcn_readfirstlane(enc_lds_addr(smem) + (uint32_t)((wb % FQA_WSLOTS) * FQA_BLOCK_BYTES + wave * FQA_PPW * 1024));
#pragma unroll
        for (int i = 0; i < FQA_PPW; ++i) enc_glds16_s(src + i * 64, (uint32_t)lane * 16u, dst + i * 1024);
    };
    issue(0);
    issue(1);
    uint4 *out = ctx + (size_t)tt * (NFB * 2 * 64) + lane;
    uint4 qq[2];
    for (int hd = 0; hd < NH; ++hd) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // 0 = K, 1 = V, 2 = Q
            const int wb = 3 * hd + j;
            // block wb has landed: younger than its pieces are block wb + 1's and - at a head's first block, in a wave that
            // has a tile - the previous head's two context stores
            if (wb == 3 * NH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (j == 0 && hd > 0 && live) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FQA_PPW + 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FQA_PPW) : "memory");
            if (j == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's K / V reads of the previous head are done)
            __builtin_amdgcn_s_barrier();  // block wb is in LDS for everyone; everyone has left block wb - 1's slot (and, j = 0, the K / V of the previous head)
            if (wb + 2 < 3 * NH) issue(wb + 2);
            if (!live) continue;
            const uint4 *w = reinterpret_cast<const uint4 *>(smem + (size_t)(wb % FQA_WSLOTS) * FQA_BLOCK_BYTES) + lane;
            f32x16 acc = {0};
            // one fragment read per MFMA (one tile per wave: qkv_kernel's two tiles share theirs): FQA_PF reads in flight cover
            // the LDS latency behind 32-cycle MFMAs
            constexpr int PF = FQA_PF;
            uint4 fr[PF + 1];
#pragma unroll
            for (int i = 0; i < PF; ++i) fr[i] = w[i * 64];
            constexpr int KSN = (FQA_ABL & 2) ? KS_H / 3 : KS_H;
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                if (ks + PF < KSN) fr[(ks + PF) % (PF + 1)] = w[(ks + PF) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc = j == 1 ? mfma(x[ks], fr[ks % (PF + 1)], acc) : mfma(fr[ks % (PF + 1)], x[ks], acc);  // V: x W (rows = tokens); Q, K: W^T x^T
            }
            const float *b = bias + block_tile(wb) * 32;
            if (j == 1) {
                const float bv = b[lane & 31];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += bv;
            } else {
                const float qs = j == 2 ? kQScaleLog2e : 1.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + b[fi(r, h)]) * qs;
            }
            if (j == 2) {
                qq[0] = acc_to_frag(acc, 0);
                qq[1] = acc_to_frag(acc, 1);
            } else {
                uint4 *dst = kvs + (size_t)wave * 256 + (size_t)(2 * j) * 64 + lane;
                dst[0] = acc_to_frag(acc, 0);
                dst[64] = acc_to_frag(acc, 1);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the head's K / V of every tile of the bin are in LDS
        if (!live) continue;
        AttnState st;
        const int n_steps = (FQA_ABL & 1) ? 1 : n_kt;
        for (int kt = 0; kt < n_steps; ++kt) {
            const uint4 *kv = kvs + (size_t)(kv0 + kt) * 256 + lane;
            const uint4 k0 = kv[0], k1 = kv[64], v0 = kv[128], v1 = kv[192];
            attn_step1_rt(st, qq, k0, k1, v0, v1, kt == 0, kt == n_steps - 1, 32 * kt, seq_len, h);
        }
        attn_store(st, out + (size_t)(hd * 2) * 64);
    }
}


// Latency path, batches whose sequences all fit one token tile (queries): QKV projection and attention of one (head,
// tile) in ONE dispatch.  Three waves compute the head's Q, K and V fragments exactly as qkv_small_kernel does (same
// chains, same bias / scale arithmetic), hand them over through LDS, and the first wave runs the one attention step:
// 12 of a query's 86 dependent dispatches go, and Q / K / V never touch global memory.
// wave j (0 = Q, 1 = K, 2 = V) of a head: the fragment pair from the tile's activation fragments x, as qkv_small_kernel
__device__ __forceinline__ void qkv_single_wave(const uint4 (&x)[KS_H], const uint4 *__restrict__ wqkv, const float *__restrict__ bqkv,
                                                int j, int head, int lane, uint4 (*frag)[2][64]) {
    const int h = lane >> 5, tile = j * NH + head;
    const uint4 *wp = wqkv + (size_t)tile * (KS_H * 64) + lane;
    uint4 w[KS_H];
#pragma unroll
    for (int ks = 0; ks < KS_H; ++ks) w[ks] = wp[ks * 64];
    f32x16 acc = {0};
    const float *b = bqkv + tile * 32;
    if (j == 2) {  // V: x W (rows = tokens)
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) acc = mfma(x[ks], w[ks], acc);
        const float bv = b[lane & 31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += bv;
    } else {       // Q, K: W^T x^T (rows = head features)
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) acc = mfma(w[ks], x[ks], acc);
        const float qs = j == 0 ? kQScaleLog2e : 1.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + b[fi(r, h)]) * qs;
    }
    frag[j][0][lane] = acc_to_frag(acc, 0);
    frag[j][1][lane] = acc_to_frag(acc, 1);
}

__global__ __launch_bounds__(192) void qkv_attention_single_kernel(const uint4 *__restrict__ act, const uint4 *__restrict__ wqkv,
                                                                   const float *__restrict__ bqkv, const TileInfo *__restrict__ ti,
                                                                   uint4 *__restrict__ ctx) {
    __shared__ uint4 frag[3][2][64];
    const int lane = threadIdx.x & 63, h = lane >> 5, j = threadIdx.x >> 6;  // j: 0 = Q, 1 = K, 2 = V
    const int head = blockIdx.x, tt = blockIdx.y;
    {
        const uint4 *xin = act + (size_t)tt * (NFB * 2 * 64) + lane;
        uint4 x[KS_H];
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) x[ks] = xin[ks * 64];
        qkv_single_wave(x, wqkv, bqkv, j, head, lane, frag);
    }
    __syncthreads();
    if (j != 0) return;
    AttnState st;
    attn_step(st, frag[0][0][lane], frag[0][1][lane], frag[1][0][lane], frag[1][1][lane], frag[2][0][lane], frag[2][1][lane], true, true, 0,
              ti[tt].seq_len, h);
    attn_store(st, ctx + (size_t)tt * (NFB * 2 * 64) + (size_t)(head * 2) * 64 + lane);
}

// The same with the PREVIOUS layer's closing LayerNorm in front (layers 1 .. 11 of the single-tile path): the workgroup's
// four waves compute the tile's LayerNorm into LDS (each of a tile's 12 workgroups repeats it; workgroup 0 writes it out
// for the later residual), waves 0-2 project from those fragments, wave 0 attends.  `ctx` must not be the residual's
// buffer (other workgroups of the tile may still be reading it).
__global__ __launch_bounds__(256) void ln_qkv_attention_single_kernel(const float *__restrict__ Y, const uint4 *__restrict__ resid,
                                                                      const float *__restrict__ bias, const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta, uint4 *__restrict__ act_out,
                                                                      const uint4 *__restrict__ wqkv, const float *__restrict__ bqkv,
                                                                      const TileInfo *__restrict__ ti, uint4 *__restrict__ ctx) {
    __shared__ float xs[2][4][64];
    __shared__ uint4 xf[NFB * 2 * 64];  // the tile's LayerNorm output, ACT fragments (24 KiB)
    __shared__ uint4 frag[3][2][64];
    const int lane = threadIdx.x & 63, h = lane >> 5, w = threadIdx.x >> 6;
    const int head = blockIdx.x, tt = blockIdx.y;
    ln4_tile(Y, tt, resid, bias, gamma, beta, xf, lane, w, xs);
    __syncthreads();
    if (blockIdx.x == 0) {
        uint4 *out = act_out + (size_t)tt * (NFB * 2 * 64);
        for (int i = threadIdx.x; i < NFB * 2 * 64; i += 256) out[i] = xf[i];
    }
    if (w < 3) {
        uint4 x[KS_H];
#pragma unroll
        for (int ks = 0; ks < KS_H; ++ks) x[ks] = xf[ks * 64 + lane];
        qkv_single_wave(x, wqkv, bqkv, w, head, lane, frag);
    }
    __syncthreads();
    if (w != 0) return;
    AttnState st;
    attn_step(st, frag[0][0][lane], frag[0][1][lane], frag[1][0][lane], frag[1][1][lane], frag[2][0][lane], frag[2][1][lane], true, true, 0,
              ti[tt].seq_len, h);
    attn_store(st, ctx + (size_t)tt * (NFB * 2 * 64) + (size_t)(head * 2) * 64 + lane);
}

int32_t launch_qkv_attention_single(const uint4 *act, const uint4 *wqkv, const float *bqkv, const TileInfo *ti, int n_tiles,
                                    uint4 *ctx, hipStream_t stream) {
    qkv_attention_single_kernel<<<dim3(NH, n_tiles), dim3(192), 0, stream>>>(act, wqkv, bqkv, ti, ctx);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

int32_t launch_ln_qkv_attention_single(const float *Y, const uint4 *resid, const float *bias, const float *gamma, const float *beta,
                                       uint4 *act_out, const uint4 *wqkv, const float *bqkv, const TileInfo *ti, int n_tiles,
                                       uint4 *ctx, hipStream_t stream) {
    ln_qkv_attention_single_kernel<<<dim3(NH, n_tiles), dim3(256), 0, stream>>>(Y, resid, bias, gamma, beta, act_out, wqkv, bqkv, ti, ctx);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

int32_t launch_attention2(const uint4 *qf, const uint4 *kf, const uint4 *vf, const int32_t *units, int n_units, uint4 *ctx,
                          hipStream_t stream) {
    const int n_wg = (n_units * NH + 3) / 4;
    attention2_kernel<<<dim3((n_wg + 7) / 8 * 8), dim3(256), 0, stream>>>(qf, kf, vf, reinterpret_cast<const int4 *>(units), n_units, ctx);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

int32_t fqa_prepare() {
    auto kern = fused_qkv_attention_kernel;
    MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FQA_LDS_BYTES));
    return MIR_OK;
}
// bins: n_bins x FQA_TILES slots {tile or -1, first slot of the sequence | key tiles << 8, sequence length, 0} (encoder.hip: build_batch)
int32_t launch_fused_qkv_attention(const uint4 *act, const uint4 *wqkv, const float *bqkv, const int32_t *bins, int n_bins, uint4 *ctx,
                                   hipStream_t stream) {
    if (n_bins <= 0) return MIR_OK;
    fused_qkv_attention_kernel<<<dim3(n_bins), dim3(64 * FQA_TILES), FQA_LDS_BYTES, stream>>>(act, wqkv, bqkv, reinterpret_cast<const int4 *>(bins), ctx);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

int32_t launch_attention(const uint4 *qf, const uint4 *kf, const uint4 *vf, const TileInfo *ti, int n_tiles,
                         uint4 *ctx, hipStream_t stream) {
    const int n_wg = (n_tiles * NH + 3) / 4;
    attention_kernel<<<dim3((n_wg + 7) / 8 * 8), dim3(256), 0, stream>>>(qf, kf, vf, ti, n_tiles, ctx);  // a multiple of 8: see the XCD mapping
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

}  // namespace enc
}  // namespace mir
