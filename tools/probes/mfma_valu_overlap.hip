// Probe: do MFMA and VALU work of two different waves on the same SIMD overlap?
// 512 threads = 8 waves = 2 per SIMD.  mode 0: every wave MFMA; 1: every wave VALU; 2: waves 0-3 MFMA,
// waves 4-7 VALU (one of each per SIMD); 3: only waves 0-3 MFMA (others exit); 4: only waves 4-7 VALU;
// 5: like 2 but VALU = v_exp_f32; 6: like 2 with s_setprio 1 on the VALU waves; 7: like 2 with s_setprio 1 on
// the MFMA waves; 8: like 2 with the roles swapped (waves 0-3 VALU, 4-7 MFMA); 9 / 10 / 11: like 2 with 24 / 16 / 28
// cycles of s_nop after every MFMA (the MFMA wave paces itself instead of queueing at the issue port).   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int PACE_CYC>
__global__ __launch_bounds__(512) void k(float* out, long long* dur, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool lo = mode == 8 ? wave >= 4 : wave < 4;
    constexpr int pace = PACE_CYC;
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3 || mode >= 5) && lo);
    const bool do_valu = mode == 1 || ((mode == 2 || mode == 4 || mode >= 5) && !lo);
    if ((mode == 6 && do_valu) || (mode == 7 && do_mfma)) __builtin_amdgcn_s_setprio(1);
    float r = 0.f;
    const long long t_begin = wall_clock64();
    if (do_mfma) {
        f32x16_t a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        bf16x8_t x = {1, 1, 1, 1, 1, 1, 1, 1}, y = x;
#define PACE()                                                                           \
    if constexpr (pace == 24) { asm volatile("s_nop 15\n\ts_nop 7"); }                                      \
    else if constexpr (pace == 16) { asm volatile("s_nop 15"); }                                            \
    else if constexpr (pace == 28) { asm volatile("s_nop 15\n\ts_nop 11"); }
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0); PACE()
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0); PACE()
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0); PACE()
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0); PACE()
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (do_valu) {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.001f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (mode == 5) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.5f;
                else v[j] = fmaf(v[j], 1.0001f, 0.5f);
            }
        }
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    const long long t_end = wall_clock64();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) dur[wave] = (do_mfma || do_valu) ? (t_end - t_begin) : 0;
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    long long* dd;
    hipMalloc(&dd, 8 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode <= 11; ++mode) {
        auto launch = [&]() {
            if (mode == 9) hipLaunchKernelGGL(k<24>, dim3(256), dim3(512), 0, 0, d, dd, mode, iters);
            else if (mode == 10) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, d, dd, mode, iters);
            else if (mode == 11) hipLaunchKernelGGL(k<28>, dim3(256), dim3(512), 0, 0, d, dd, mode, iters);
            else hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, dd, mode, iters);
        };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[8];
        hipMemcpy(h, dd, 64, hipMemcpyDeviceToHost);
        // wall_clock64 ticks at 100 MHz
        printf("mode %2d: %.3f ms kernel | block 0: waves 0-3 %.3f ms, waves 4-7 %.3f ms\n", mode, ms, h[0] * 1e-5, h[4] * 1e-5);
    }
    return 0;
}
