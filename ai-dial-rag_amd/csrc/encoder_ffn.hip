// FFN of the bge-small-en encoder: FFN1 + GELU + FFN2 + residual + LayerNorm in one kernel - see encoder_common.h
// for the layouts.  Own translation unit, built with -mllvm -amdgpu-mfma-vgpr-form=1 (accumulators in VGPRs).
//
// Workgroup = 8 waves = one GROUP of 128 tokens at a time (one persistent workgroup per CU walks the groups, the pipeline
// below does not drain between them: encoder_ffn_kernel.h), TWO waves per SIMD with different ROLES on the same 32-token tile:
//   role A (waves 0-3)  h^T(ht) = gelu(W1(ht)^T x^T + b1): the tile's activations x stay in registers (96 VGPRs) as the B
//                       operand, W1 streams through LDS; 24 MFMAs + 16 GELUs (~290 VALU) per intermediate tile ht;
//                       hands h(ht) (2 KiB of float16 fragments) to its partner through LDS;
//   role B (waves 4-7)  y^T += W2(ht)^T h^T(ht): all 12 output tiles accumulate in registers (192 VGPRs), 24 MFMAs per
//                       ht; moves the weight stream (LDS-DMA, no registers); bias + residual + LayerNorm in its epilogue.
// Wave w and wave w + 4 land on the same SIMD (waves go to SIMDs cyclically), and both roles fit two-per-SIMD (<= 256
// registers) because neither holds x AND y.
//
// Weight stream: stage j (48 KiB) = [ W1(j+1) | W2(j-1) ] - what A and B need between barrier j and barrier j+1 (B runs one
// tile behind A: it consumes h(j-1), which A wrote before barrier j).  Two LDS slots; at the top of stage j the B waves
// issue the 48 LDS-DMA pieces of stage j+1 into the free slot and wait for them before the next barrier.
//
// What bounds it (in-kernel stamps and ablations, tools/ffn_stamps.hip, cycles per stage of 48 MFMAs = 1536 cycles of
// matrix pipe per SIMD): the weight stream alone, nothing computed, takes 1.7-1.8K cycles per stage whichever way it is
// moved (plain loads -> registers -> ds_write by 4 waves: 1685; LDS-DMA by 4 waves: 1835) = 26-29 B/clk per CU of
// L2-resident data (TCC hit rate 87 %: the weights do stay in L2) against the 32 B/clk the matrix pipe would need at
// this tiling (128 tokens per 48 KiB of weights; x and y resident in registers leave room for no more tokens per CU);
// role A alone, weights in place: 2.3K (0.9K of it the exact-erf GELU: one rcp, one exp and ~16 other VALU per value);
// role B alone: 1.0K; everything together 3.3K (3.9K with the A waves also moving the weights through registers, the
// round-1 single-role kernel 4.0-4.3K).  So: a memory-pipe floor next to the MFMA time, plus a serial GELU chain that the
// partner wave cannot take over without also holding x.
#include <algorithm>
#include "common.h"
#include "encoder_common.h"
#include "encoder_ffn_kernel.h"

namespace mir {
namespace enc {

int32_t ffn_prepare() {
    auto kern = ffn_ln_kernel<false>;
    MIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS_BYTES));
    return MIR_OK;
}

int32_t launch_ffn(const uint4 *act_in, int n_tiles, const unsigned char *wffn, const float *params, uint4 *act_out,
                   hipStream_t stream) {
    ffn_ln_kernel<false><<<dim3(std::min((n_tiles + 3) / 4, FFN_MAX_GRID)), dim3(512), FFN_LDS_BYTES, stream>>>(act_in, n_tiles, wffn, params, act_out, nullptr);
    MIR_HIP(hipGetLastError());
    return MIR_OK;
}

}  // namespace enc
}  // namespace mir
