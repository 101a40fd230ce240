// Probe: two waves per SIMD in anti-phase - one in a cluster of NM MFMAs (optionally paced: s_nop after each MFMA so that it
// does not queue at the issue port), the other in a VALU cluster (NE v_exp + NF v_fma), roles swap at an s_barrier (8-wave
// workgroup, waves w and w + 4 share a SIMD).  Reports ns per interval against the MFMA cluster alone.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pingpong_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int MODE, int PACE, int NE, int NF>  // MODE 0: ping-pong, 1: MFMA clusters only (both groups, alternating), 2: every wave does M then V (no roles)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    f32x16_t a[4] = {{0}, {0}, {0}, {0}};
    bf16x8_t x = {1, 1, 1, 1, 1, 1, 1, 1}, y = x;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.001f + j;
    auto mcluster = [&]() {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            a[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a[m & 3], 0, 0, 0);
            if (PACE == 16) asm volatile("s_nop 15");
            if (PACE == 20) asm volatile("s_nop 15\n\ts_nop 3");
            if (PACE == 24) asm volatile("s_nop 15\n\ts_nop 7");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto vcluster = [&]() {
#pragma unroll
        for (int j = 0; j < NE; ++j) v[j & 15] = __builtin_amdgcn_exp2f(v[j & 15]);
#pragma unroll
        for (int j = 0; j < NF; ++j) v[j & 15] = fmaf(v[j & 15], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            if (grp == 0) mcluster(); else vcluster();
            __builtin_amdgcn_s_barrier();
            if (grp == 1) mcluster(); else vcluster();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 1) {
            if (grp == 0) mcluster();
            __builtin_amdgcn_s_barrier();
            if (grp == 1) mcluster();
            __builtin_amdgcn_s_barrier();
        } else {
            mcluster(); vcluster();
            __builtin_amdgcn_s_barrier();
        }
    }
    float r = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int j = 0; j < 16; ++j) r += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE, int PACE, int NE, int NF>
float run(float* d) {
    const int iters = 5000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PACE, NE, NF>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PACE, NE, NF>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;  // ns per iteration = 64 MFMAs per SIMD + 2 VALU clusters
}
template <int NE, int NF>
void row(float* d) {
    printf("VALU cluster %d exp + %d fma: MFMA only %.0f ns | ping-pong unpaced %.0f | paced 16: %.0f  20: %.0f  24: %.0f | no roles (M then V in every wave) %.0f\n", NE, NF,
           run<1, 0, NE, NF>(d), run<0, 0, NE, NF>(d), run<0, 16, NE, NF>(d), run<0, 20, NE, NF>(d), run<0, 24, NE, NF>(d), run<2, 0, NE, NF>(d));
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    row<32, 80>(d);
    row<32, 160>(d);
    row<0, 200>(d);
    return 0;
}
