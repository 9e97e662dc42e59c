// Attention kernel of the bge-small-en encoder - see encoder_common.h for the layouts and why this
// kernel has a translation unit (and one compiler option) of its own.
#include "common.h"
#include "encoder_common.h"

namespace mir {
namespace enc {

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
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_tiles * NH) return;
    const int head = w / n_tiles, tt = w - head * n_tiles;   // neighbours share a sequence's K/V of one head
    const TileInfo info = ti[tt];
    // Q arrives pre-multiplied by log2(e) / sqrt(32) (qkv kernels): the scores are exp2 arguments as they leave the MFMA
    const uint4 *qp = qf + ((size_t)(tt * NH + head) * 2) * 64 + lane;
    const uint4 q0 = qp[0], q1 = qp[64];
    typedef float __attribute__((ext_vector_type(2))) f32x2;
    f32x16 o = {0};
    float m = -__builtin_inff(), l = 0.f;
    for (int kt = 0; kt < info.seq_tiles; ++kt) {
        const size_t kb = ((size_t)((info.seq_first_tile + kt) * NH + head) * 2) * 64 + lane;
        const uint4 k0 = kf[kb], k1 = kf[kb + 64];
        const uint4 v0 = vf[kb], v1 = vf[kb + 64];
        f32x16 s = {0};
        s = mfma(k0, q0, s);
        s = mfma(k1, q1, s);
        if (kt == info.seq_tiles - 1) {  // only the sequence's last key tile can hold padding
            const int key0 = 32 * kt;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (key0 + fi(r, h) < info.seq_len) ? s[r] : -__builtin_inff();
        }
        const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]), a2 = fmaxf(fmaxf(s[6], s[7]), s[8]);
        const float a3 = fmaxf(fmaxf(s[9], s[10]), s[11]), a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
        const float mx = half_max(fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(a4, s[15])));
        const float m_new = fmaxf(m, mx);
        // v_exp_f32 directly: exp2f() wraps it in a compare / select / ldexp to keep results below
        // 2^-126 exact, ~4 extra instructions per value in a VALU-bound loop; a softmax term that small
        // is zero next to the row's maximum term 1.0 either way.  Subtraction and sum two values per instruction
        // (v_pk_add_f32): the loop is VALU-bound 4:1 against its MFMAs.
        const f32x2 nm = {-m_new, -m_new};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f32x2 d = f32x2{s[r], s[r + 1]} + nm;
            d.x = __builtin_amdgcn_exp2f(d.x);
            d.y = __builtin_amdgcn_exp2f(d.y);
            s[r] = d.x;
            s[r + 1] = d.y;
            ps2 += d;
        }
        const float ps = half_sum(ps2.x + ps2.y);
        if (__any(m_new > m)) {  // some query's running maximum moved: rescale (wave-uniform branch)
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);  // exp2(-inf) = 0 on the first tile
            l = fmaf(l, alpha, ps);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
        } else {
            l += ps;
        }
        m = m_new;
        // P to float16 with the packed convert (two values per instruction; the scalar casts cost three)
        const uint4 p0 = make_uint4(pack2_rn(s[0], s[1]), pack2_rn(s[2], s[3]), pack2_rn(s[4], s[5]), pack2_rn(s[6], s[7]));
        const uint4 p1 = make_uint4(pack2_rn(s[8], s[9]), pack2_rn(s[10], s[11]), pack2_rn(s[12], s[13]), pack2_rn(s[14], s[15]));
        o = mfma(v0, p0, o);
        o = mfma(v1, p1, o);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= inv;
    uint4 *out = ctx + (size_t)tt * (NFB * 2 * 64) + (size_t)(head * 2) * 64 + lane;
    out[0] = acc_to_frag(o, 0);
    out[64] = acc_to_frag(o, 1);
}


int32_t launch_attention(const uint4 *qf, const uint4 *kf, const uint4 *vf, const TileInfo *ti, int n_tiles,
                         uint4 *ctx, hipStream_t stream) {
    attention_kernel<<<dim3((n_tiles * NH + 3) / 4), dim3(256), 0, stream>>>(qf, kf, vf, ti, n_tiles, ctx);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

}  // namespace enc
}  // namespace mir
