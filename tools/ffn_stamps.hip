// Diagnostic build of the encoder's FFN kernel with in-kernel time stamps (never part of the library):
//   hipcc -O3 -std=c++17 -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -I ai-dial-rag_amd/csrc tools/ffn_stamps.hip -o /tmp/ffn_stamps
// Random weights / activations of a full 3072-tile pass; prints, for workgroup 0, the cycles of each stage per wave
// (barrier exit -> end of the wave's stage work -> next barrier exit), for the workgroup's FIRST group (the kernel is
// persistent: 256 workgroups walk the groups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "common.h"
namespace mir { void set_error(const char *, ...) {} }
#include "encoder_ffn_kernel.h"
using namespace mir::enc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int n_tiles = argc > 1 ? atoi(argv[1]) : 3072;
    const size_t act_bytes = (size_t)n_tiles * NFB * 2 * 64 * 16, w_bytes = (size_t)NHT * FFN_STAGE_BYTES;
    std::vector<_Float16> hact(act_bytes / 2), hw(w_bytes / 2);
    for (auto &v : hact) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    for (auto &v : hw) v = (_Float16)((rand() % 2001 - 1000) / 20000.0f);
    std::vector<float> hp(FFN_PARAM_FLOATS + GELU_LUT_FLOATS, 0.01f);
    gelu_table(hp.data() + FFN_PARAM_FLOATS);
    for (int i = FF + H; i < FF + 2 * H; ++i) hp[i] = 1.0f;
    void *act, *out, *w, *p; unsigned long long *st;
    CK(hipMalloc(&act, act_bytes)); CK(hipMalloc(&out, act_bytes)); CK(hipMalloc(&w, w_bytes)); CK(hipMalloc(&p, hp.size() * 4));
    CK(hipMalloc((void **)&st, 8 * (NHT + 1) * 2 * 8));
    CK(hipMemcpy(act, hact.data(), act_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), w_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    auto kern = ffn_ln_kernel<true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 40; ++rep) {
        if (rep == 20) CK(hipEventRecord(e0));
        kern<<<dim3(std::min((n_tiles + 3) / 4, FFN_MAX_GRID)), dim3(512), FFN_LDS_BYTES>>>((const uint4 *)act, n_tiles, (const unsigned char *)w, (const float *)p, (uint4 *)out, st);
    }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d tiles: %.1f us per launch\n", n_tiles, ms / 20 * 1e3);
    std::vector<unsigned long long> hs(8 * (NHT + 1) * 2);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    auto at = [&](int wv, int s, int k) { return hs[(wv * (NHT + 1) + s) * 2 + k]; };
    printf("stage: A(wave0) work, A wait | B(wave4) work, B wait   [cycles]\n");
    for (int s = 1; s < NHT; s += 6)
        printf("%3d: %6llu %6llu | %6llu %6llu\n", s, at(0, s, 1) - at(0, s, 0), at(0, s + 1, 0) - at(0, s, 1), at(4, s, 1) - at(4, s, 0), at(4, s + 1, 0) - at(4, s, 1));
    printf("whole loop: %llu cycles = %llu per stage\n", at(0, NHT, 0) - at(0, 0, 0), (at(0, NHT, 0) - at(0, 0, 0)) / NHT);
    return 0;
}
