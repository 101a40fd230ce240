// Probe: the slab GEMM's step structure (8 waves, 2 per SIMD; per step and wave 24 ds_read_b128 of operand
// fragments + 1024 cycles of MFMA passes, then a workgroup barrier) with the MFMAs issued as
//   mode 0: 64 x v_mfma_f32_16x16x32_bf16 (what the kernel does: 12 reads -> 32 MFMAs, twice)
//   mode 1: 32 x v_mfma_f32_32x32x16_bf16 (same FLOPs, same LDS bytes: 6 reads -> 8 MFMAs, four times)
//   mode 2/3: as 0/1 without the LDS reads (MFMA + barrier only)
//   mode 4: mode 0 + every wave issues 4 LDS-DMA instructions (1 KiB each, from L2) per step, drained before the barrier
//   mode 5: mode 0 + waves 0-3 issue 8 LDS-DMA instructions each, waves 4-7 none
// Prints cycles per step (s_memtime) - is a step bound by the MFMA pipe (2048 cycles per SIMD) or by issue?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/step_shape_mfma.hip -o /tmp/ssm && /tmp/ssm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int steps, const unsigned char* src) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[96 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 96 * 1024 / 16; i += 512) ((uint4*)lds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    const unsigned char* base = lds + (tid >> 6) * 4096 + lane * 16;  // conflict-free: consecutive lanes, consecutive 16 B
    f32x4_t a16[4][8];
    f32x16_t a32[2][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a16[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) a32[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        const unsigned char* b = base + (s & 7) * 2048;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MODE >= 4) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 20, 0x00020000);
            const int wv = tid >> 6, nd = MODE == 4 ? 4 : (wv < 4 ? 8 : 0);
            for (int i = 0; i < nd; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 64 * 1024 + ((s & 1) * 32 + (i * (MODE == 4 ? 8 : 4) + (wv & (MODE == 4 ? 7 : 3)))) * 1024 % (32 * 1024)),
                                                         16, (unsigned)(((s * 37 + i * 8 + wv) & 511) * 1024 + lane * 16), 0, 0, 0);
        }
#endif
        if constexpr (MODE == 0 || MODE == 2 || MODE >= 4) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fw[4], fx[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) fw[i] = (MODE == 0 || MODE >= 4) ? *(const bf16x8_t*)(b + (ks * 12 + i) * 1024) : (bf16x8_t){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
                for (int i = 0; i < 8; ++i) fx[i] = (MODE == 0 || MODE >= 4) ? *(const bf16x8_t*)(b + (ks * 12 + 4 + i) * 1024) : (bf16x8_t){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) a16[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fx[mi], a16[ni][mi], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8_t fw[2], fx[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) fw[i] = MODE == 1 ? *(const bf16x8_t*)(b + (ks * 6 + i) * 1024) : (bf16x8_t){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
                for (int i = 0; i < 4; ++i) fx[i] = MODE == 1 ? *(const bf16x8_t*)(b + (ks * 6 + 2 + i) * 1024) : (bf16x8_t){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) a32[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ni], fx[mi], a32[ni][mi], 0, 0, 0);
            }
        }
        if constexpr (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r += a16[i][j][0];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) r += a32[i][j][0];
    out[blockIdx.x * 512 + tid] = r;
    if (blockIdx.x == 128 && lane == 0) cyc[tid >> 6] = t1 - t0;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    unsigned long long* c; hipMalloc(&c, 64);
    unsigned char* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0x3f, 1 << 20);
    const int steps = 2000;
    const char* names[6] = {"16x16x32 + 24 LDS reads", "32x32x16 + 24 LDS reads", "16x16x32, no LDS reads", "32x32x16, no LDS reads",
                            "16x16x32 + reads + 4 DMAs per wave", "16x16x32 + reads + 8 DMAs on waves 0-3"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, c, steps, src);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, c, steps, src);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, c, steps, src);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, d, c, steps, src);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, d, c, steps, src);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, d, c, steps, src);
            hipDeviceSynchronize();
        }
        unsigned long long h[8];
        hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.0f cycles per step (ideal 2048)\n", mode, names[mode], (double)h[0] / steps);
    }
    return 0;
}
