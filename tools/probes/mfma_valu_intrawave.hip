// Probe: does VALU work placed between the MFMAs of ONE wave hide under them?  256 threads = 1 wave per SIMD.
// mode 0: 4 MFMA 32x32x16 per iteration; mode 1: 4 x NV VALU per iteration; mode 2: 4 x (1 MFMA + NV VALU).
// NV x kind: fma (4-cycle issue) or exp (transcendental).   hipcc --offload-arch=gfx950 -O3 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int MODE, int NV, bool EXP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16_t a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8_t x = {1, 1, 1, 1, 1, 1, 1, 1}, y = x;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
    auto valu = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j % 8] = EXP ? __builtin_amdgcn_exp2f(v[j % 8]) : fmaf(v[j % 8], 1.0001f, 0.5f);
    };
    for (int i = 0; i < iters; ++i) {
        if (MODE != 1) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        if (MODE != 0) valu();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        if (MODE != 0) valu();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 1) a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        if (MODE != 0) valu();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 1) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        if (MODE != 0) valu();
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int j = 0; j < 8; ++j) r += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE, int NV, bool EXP>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV, EXP>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, EXP>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / 4;  // ns per (MFMA + NV VALU) group
}
template <int NV, bool EXP>
void row(float* d) {
    const int iters = 20000;
    printf("NV=%d %s: mfma only %.1f ns | valu only %.1f ns | interleaved %.1f ns per group\n", NV, EXP ? "exp" : "fma",
           run<0, NV, EXP>(d, iters), run<1, NV, EXP>(d, iters), run<2, NV, EXP>(d, iters));
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 4);
    row<2, false>(d); row<4, false>(d); row<6, false>(d); row<8, false>(d);
    row<1, true>(d); row<2, true>(d); row<4, true>(d);
    return 0;
}
