// Attention kernel of the bge-small-en encoder - see encoder_common.h for the layouts and why this
// kernel has a translation unit (and one compiler option) of its own.
#include "common.h"
#include "encoder_common.h"

namespace mir {
namespace enc {

// The running state of one (head, query tile) and one key tile's worth of attention: S^T = K Q^T - ref, P = exp2,
// O^T += V^T P^T, lsum += 1 P^T (the lazy softmax reference is described at attention_kernel).  Shared by the throughput /
// general kernel and the single-tile fused kernel of the latency path, so both round alike.
struct AttnState {
    f32x16 o = {0}, lsum = {0}, nref = {0};  // lsum: every register holds the lane's query's running sum
};
__device__ __forceinline__ void attn_step(AttnState &st, const uint4 &q0, const uint4 &q1, const uint4 &kc0, const uint4 &kc1,
                                          const uint4 &vc0, const uint4 &vc1, bool first, bool last, int key0, int seq_len, int h) {
    constexpr float kSlack = 6.0f;
    const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
    f32x16 s = mfma(kc0, q0, st.nref);
    s = mfma(kc1, q1, s);
    if (last) {  // only the sequence's last key tile can hold padding
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (key0 + fi(r, h) < seq_len) ? s[r] : -__builtin_inff();
    }
    const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]), a2 = fmaxf(fmaxf(s[6], s[7]), s[8]);
    const float a3 = fmaxf(fmaxf(s[9], s[10]), s[11]), a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
    const float mx = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(a4, s[15]));  // this lane's half of the keys
    if (first || __any(mx > kSlack)) {  // wave-uniform; rare after the first tile
        const float mq = half_max(mx);
        const float delta = first ? mq : fmaxf(mq, 0.f);  // the reference only rises after the first tile
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] -= delta;
            st.nref[r] -= delta;
            st.o[r] *= alpha;
            st.lsum[r] *= alpha;
        }
    }
    // v_exp_f32 directly: exp2f() wraps it in a compare / select / ldexp to keep results below 2^-126 exact, ~4 extra
    // instructions per value; a softmax term that small is zero next to the row's largest term either way
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
    // P to float16 with the packed convert (two values per instruction; the scalar casts cost three)
    const uint4 p0 = make_uint4(pack2_rn(s[0], s[1]), pack2_rn(s[2], s[3]), pack2_rn(s[4], s[5]), pack2_rn(s[6], s[7]));
    const uint4 p1 = make_uint4(pack2_rn(s[8], s[9]), pack2_rn(s[10], s[11]), pack2_rn(s[12], s[13]), pack2_rn(s[14], s[15]));
    st.o = mfma(vc0, p0, st.o);
    st.lsum = mfma(ones, p0, st.lsum);
    st.o = mfma(vc1, p1, st.o);
    st.lsum = mfma(ones, p1, st.lsum);
}
// normalise and write the context fragments of the (head, query tile)
__device__ __forceinline__ void attn_store(AttnState &st, uint4 *__restrict__ out) {
    const float inv = 1.0f / st.lsum[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) st.o[r] *= inv;
    enc_store_nt(out, acc_to_frag(st.o, 0));  // the context is read once, by the output projection (measured: within noise
    enc_store_nt(out + 64, acc_to_frag(st.o, 1));  // for this kernel, 361 vs 364 us; the projection after it 197 vs 202 us)
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
    auto step = [&](int kt, const uint4 &kc0, const uint4 &kc1, const uint4 &vc0, const uint4 &vc1) {
        attn_step(st, q0, q1, kc0, kc1, vc0, vc1, kt == 0, kt == n_kt - 1, 32 * kt, seq_len, h);
    };
    // Two key tiles per trip, each in its own registers: a tile's K/V are requested one step ahead.  vmcnt counts in
    // order, so the wait for the older buffer leaves the younger one's loads in flight.  (The requests are unconditional -
    // past the end they re-read the last tile: behind a branch that may or may not have issued loads, hipcc waits for
    // vmcnt(0).)
    auto load = [&](int kt, uint4 &k0, uint4 &k1, uint4 &v0, uint4 &v1) {
        const size_t kb = kv0 + (size_t)kt * (NH * 2 * 64);
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

int32_t launch_attention(const uint4 *qf, const uint4 *kf, const uint4 *vf, const TileInfo *ti, int n_tiles,
                         uint4 *ctx, hipStream_t stream) {
    const int n_wg = (n_tiles * NH + 3) / 4;
    attention_kernel<<<dim3((n_wg + 7) / 8 * 8), dim3(256), 0, stream>>>(qf, kf, vf, ti, n_tiles, ctx);  // a multiple of 8: see the XCD mapping
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

}  // namespace enc
}  // namespace mir
