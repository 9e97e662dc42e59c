// Is v_exp_f32's rate per SIMD or per CU?  (round 4: every form of the attention kernel ends at ~600 cycles per (query tile,
// key tile) step and SIMD whatever is done to its loads, its other vector work or its occupancy; 16 v_exp_f32 per step would
// explain that if the four SIMDs of a CU shared one transcendental pipe.)  N dependent-free v_exp_f32 (or v_fma_f32) per wave,
// 1 / 4 / 8 / 16 waves per CU, one workgroup per CU.   hipcc --offload-arch=gfx950 -O3 tools/exp_rate.hip -o /tmp/exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void rate_kernel(float *out, int iters, float seed) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed + 0.001f * (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
            else asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
static void run(const char *name, int waves_per_cu, int cus) {
    float *out;
    (void)hipMalloc(&out, 64);
    const int iters = 20000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    rate_kernel<OP><<<cus, 64 * waves_per_cu>>>(out, 1000, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    rate_kernel<OP><<<cus, 64 * waves_per_cu>>>(out, iters, 0.5f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const double per_wave_ns = ms * 1e6 / ((double)iters * 8);
    printf("%-10s waves/CU %2d: %.3f ms  -> %.2f ns per instruction per wave  (%.2f ns per instruction per CU)\n", name, waves_per_cu, ms,
           per_wave_ns, per_wave_ns / waves_per_cu);
    (void)hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, cus);
    for (int w : {1, 4, 8, 16}) run<0>("v_exp_f32", w, cus);
    for (int w : {1, 4, 8, 16}) run<2>("v_rcp_f32", w, cus);
    for (int w : {1, 4, 8, 16}) run<1>("v_fma_f32", w, cus);
    return 0;
}
