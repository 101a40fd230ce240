// Probe: what does the chip sustain on WRITES alone, on reads alone, and on a copy?  (The GEMM launches whose output is three
// times their input - the MHA in-projection - run at the same ~34 us on two quite different kernels.)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_write_bw tools/probes/hbm_write_bw.hip && /tmp/hbm_write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float v4f;
template <int MODE>  // 0 fill, 1 fill non-temporal, 2 read (sum), 3 copy, 4 copy with non-temporal stores; 5 / 6: see k2
__global__ __launch_bounds__(256) void k(float4* dst, const float4* src, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
        if (MODE == 1) __builtin_nontemporal_store((v4f){1.f, 2.f, 3.f, (float)i}, (v4f*)dst + i);
        if (MODE == 2) { const float4 v = src[i]; acc += v.x + v.w; }
        if (MODE == 3) dst[i] = src[i];
        if (MODE == 4) __builtin_nontemporal_store(((const v4f*)src)[i], (v4f*)dst + i);
    }
    if (MODE == 2 && acc == 1.2345f) *sink = acc;
}
// The slab GEMM's store pattern on a (rows, 768) bf16 tensor: a wave owns 16 rows x 128 bytes; MODE 5: two instructions of 16 rows x
// 64 bytes (lane = fg*16 + row: four lanes per row), the halves of a line back to back; MODE 6: two instructions of 8 rows x 128
// bytes (eight lanes per row: whole lines).
template <int MODE>
__global__ __launch_bounds__(256) void k2(unsigned char* dst, size_t rows, int rowb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nblk = rows / 16 * (rowb / 128);  // 16-row x 128-byte blocks
    for (size_t b = ((size_t)blockIdx.x * 4 + wave); b < nblk; b += (size_t)gridDim.x * 4) {
        const size_t r0 = (b / (rowb / 128)) * 16;
        unsigned char* base = dst + r0 * rowb + (b % (rowb / 128)) * 128;
        const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
        if (MODE == 5) {
            const int fr = lane & 15, fg = lane >> 4;
            *(float4*)(base + (size_t)fr * rowb + fg * 16) = v;
            *(float4*)(base + (size_t)fr * rowb + 64 + fg * 16) = v;
        } else {
            const int r = lane >> 3, c = lane & 7;
            *(float4*)(base + (size_t)r * rowb + c * 16) = v;
            *(float4*)(base + (size_t)(8 + r) * rowb + c * 16) = v;
        }
    }
}
template <int MODE> void run2(unsigned char* d, size_t rows, int rowb, const char* name) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k2<MODE>), dim3(256 * 8), dim3(256), 0, 0, d, rows, rowb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k2<MODE>), dim3(256 * 8), dim3(256), 0, 0, d, rows, rowb);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / 10, moved = (double)rows * rowb;
    printf("%-44s %6.0f MB: %7.1f us  %5.2f TB/s\n", name, moved / 1e6, t * 1e6, moved / t / 1e12);
}
template <int MODE> void run(float4* d, float4* s, size_t bytes, float* sink, const char* name) {
    const size_t n = bytes / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE>), dim3(256 * 8), dim3(256), 0, 0, d, s, n, sink);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256 * 8), dim3(256), 0, 0, d, s, n, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / 10, moved = (MODE >= 3 ? 2.0 : 1.0) * bytes;
    printf("%-28s %6.0f MB: %7.1f us  %5.2f TB/s\n", name, bytes / 1e6, t * 1e6, moved / t / 1e12);
}
int main() {
    float4 *d, *s; float* sink;
    (void)hipMalloc(&d, 1 << 30); (void)hipMalloc(&s, 1 << 30); (void)hipMalloc(&sink, 4);
    (void)hipMemset(s, 0, 1 << 30);
    for (size_t mb : {75, 1024}) {
        const size_t b = mb << 20;
        run<0>(d, s, b, sink, "fill"); run<1>(d, s, b, sink, "fill, non-temporal"); run<2>(d, s, b, sink, "read");
        run<3>(d, s, b, sink, "copy (read + write bytes)"); run<4>(d, s, b, sink, "copy, non-temporal stores");
    }
    run2<5>((unsigned char*)d, 49152, 1536, "GEMM pattern, 16 rows x 64 B per instruction");
    run2<6>((unsigned char*)d, 49152, 1536, "whole lines, 8 rows x 128 B per instruction");
    run2<5>((unsigned char*)d, 49152, 512, "GEMM pattern, N = 256");
    run2<6>((unsigned char*)d, 49152, 512, "whole lines, N = 256");
    return 0;
}
