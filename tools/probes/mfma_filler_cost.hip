// Probe: what does each kind of filler cost between the MFMAs of ONE wave per SIMD (the pipelined attention kernel's regime)?
// 256 workgroups x 256 threads, every wave: ITER x 8 gaps of [1 MFMA 32x32x16 bf16 + fillers of a kind], asm volatile throughout
// (nothing is reordered).  Prints ns and cycles (s_memtime) per MFMA gap.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfc ... && /tmp/mfc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define ADD(d, s) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(s))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define FMA(d, s) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(s))
#define MAX3(d, x, y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
#define DSR(d, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr))
#define NOP() asm volatile("s_nop 0")

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    f32x16_t acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    u32x4_t a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a, frag = a;
    float s0 = threadIdx.x * 1e-3f, s1 = s0 + 0.5f, e0 = 0, e1 = 0, h0 = 0, h1 = 0, m = 0;
    unsigned pk = 0;
    const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    for (int i = threadIdx.x; i < 16384; i += 256) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            MFMA(acc[g & 3]);
            if (KIND == 1) { EXP(e0, s0); EXP(e1, s1); }
            if (KIND == 2) { ADD(h0, e0); ADD(h1, e1); }
            if (KIND == 3) { ADD(h0, e0); ADD(h1, e1); CVT(pk, e0, e1); }
            if (KIND == 4) { EXP(e0, s0); EXP(e1, s1); ADD(h0, e0); ADD(h1, e1); CVT(pk, e0, e1); }
            if (KIND == 5) { if (g & 1) { ADD(h0, e0); ADD(h1, e1); CVT(pk, e0, e1); } else { EXP(e0, s0); EXP(e1, s1); } }
            if (KIND == 6) { FMA(h0, s0); FMA(h1, s1); FMA(e0, s0); FMA(e1, s1); }
            if (KIND == 7) { if (g & 1) { ADD(h0, e0); ADD(h1, e1); CVT(pk, e0, e1); } else { DSR(frag, addr); EXP(e0, s0); EXP(e1, s1); } }
            if (KIND == 8) { DSR(frag, addr); }
            if (KIND == 9) { EXP(e0, s0); }
            if (KIND == 10) { FMA(h0, s0); }
            if (KIND == 11) { FMA(h0, s0); FMA(h1, s1); }
            if (KIND == 12) { FMA(h0, s0); FMA(h1, s1); FMA(e0, s0); FMA(e1, s1); FMA(m, s1); FMA(s0, s1); }
            if (KIND == 13) { NOP(); NOP(); NOP(); NOP(); }
            if (KIND == 14) { MAX3(m, s0, s1); MAX3(h0, s0, s1); }
            if (KIND == 15) { if (g & 1) { ADD(h0, e0); ADD(h1, e1); CVT(pk, e0, e1); } else { EXP(e0, s0); NOP(); EXP(e1, s1); } }
        }
        if (KIND == 7 || KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = e0 + e1 + h0 + h1 + m + __uint_as_float(pk) + __uint_as_float(frag[0]);
    for (int j = 0; j < 4; ++j) r += acc[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(float* d, long long* dc, const char* what) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, d, dc, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; long long c;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("kind %2d  %6.2f ns  %6.1f memtime ticks per MFMA gap   %s\n", KIND, ms * 1e6f / iters / 8, (double)c / iters / 8, what);
}
int main() {
    float* d; long long* dc;
    (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&dc, 8);
    run<0>(d, dc, "MFMA only");
    run<10>(d, dc, "+ 1 fma");
    run<11>(d, dc, "+ 2 fma");
    run<6>(d, dc, "+ 4 fma");
    run<12>(d, dc, "+ 6 fma");
    run<13>(d, dc, "+ 4 s_nop");
    run<9>(d, dc, "+ 1 exp");
    run<1>(d, dc, "+ 2 exp");
    run<2>(d, dc, "+ 2 add");
    run<3>(d, dc, "+ 2 add + cvt_pk");
    run<14>(d, dc, "+ 2 max3");
    run<4>(d, dc, "+ 2 exp + 2 add + cvt_pk");
    run<5>(d, dc, "alternating [2 exp] [2 add + cvt_pk]");
    run<15>(d, dc, "alternating [exp nop exp] [2 add + cvt_pk]");
    run<8>(d, dc, "+ 1 ds_read_b128");
    run<7>(d, dc, "alternating [ds_read_b128 + 2 exp] [2 add + cvt_pk]");
    return 0;
}
