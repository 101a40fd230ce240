// Probe: issue interval of v_mfma_f32_32x32x16_bf16 as a function of how many independent accumulator chains rotate
// (1 = every MFMA accumulates into the previous one's result), with NV plain VALU between the MFMAs; one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_latency.hip -o /tmp/mcl && /tmp/mcl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NCH, int NV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters) {
    f32x16_t a[4] = {{0}, {0}, {0}, {0}};
    bf16x8_t x = {1, 1, 1, 1, 1, 1, 1, 1}, y = x;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            a[m % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a[m % NCH], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = fmaf(v[j % 8], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int j = 0; j < 8; ++j) r += v[j];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = r;
}
template <int NCH, int NV, int WAVES>
float run(float* d) {
    const int iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCH, NV, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCH, NV, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / 8;  // ns per MFMA of one wave
}
template <int NCH, int NV>
void row(float* d) {
    printf("chains=%d NV=%d: %.1f ns per MFMA with 1 wave/SIMD | %.1f ns per MFMA per wave with 2 waves/SIMD\n", NCH, NV, run<NCH, NV, 4>(d), run<NCH, NV, 8>(d));
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    row<1, 0>(d); row<2, 0>(d); row<4, 0>(d);
    row<1, 4>(d); row<2, 4>(d); row<4, 4>(d);
    row<1, 8>(d); row<2, 8>(d); row<4, 8>(d);
    return 0;
}
