// Probe: fillers between 16x16x32 bf16 MFMAs (16 cycles each) of ONE wave per SIMD - how much VALU hides per gap?  (The
// single-launch predictor's K loop uses this MFMA shape; a two-tile variant would put the LayerNorm epilogue of one tile between
// the MFMAs of the other.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/m16 tools/probes/mfma16_filler_cost.hip && /tmp/m16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(d, s) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(s))
#define MAX(d, s) asm volatile("v_max_f32 %0, %0, %1" : "+v"(d) : "v"(s))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    f32x4_t acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    u32x4_t a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float s0 = threadIdx.x * 1e-3f, s1 = s0 + 0.5f, h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    unsigned pk = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            MFMA(acc[g]);
            if (KIND >= 1) FMA(h0, s0);
            if (KIND >= 2) MAX(h1, s1);
            if (KIND >= 3) FMA(h2, s0);
            if (KIND >= 4) FMA(h3, s1);
            if (KIND >= 5) CVT(pk, h0, h1);
            if (KIND >= 6) FMA(h0, s1);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = h0 + h1 + h2 + h3 + __uint_as_float(pk);
    for (int j = 0; j < 16; ++j) r += acc[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> void run(float* d, long long* dc) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, d, dc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; long long c; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%d fillers per 16x16x32 MFMA: %6.2f ns  %5.1f cycles per gap\n", KIND, ms * 1e6f / iters / 16, (double)c / iters / 16);
}
int main() {
    float* d; long long* dc;
    (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&dc, 8);
    run<0>(d, dc); run<1>(d, dc); run<2>(d, dc); run<3>(d, dc); run<4>(d, dc); run<5>(d, dc); run<6>(d, dc);
    return 0;
}
