// Probe the lane/element semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                  // lane l -> elements 4l..4l+3 (linear)
    else if (mode == 1) addr = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;  // 4 rows x 16 cols per 16-lane group, row stride 128 el
    else addr = (l & 15) * 256 + (l >> 4) * 8;                     // each lane its own row (stride 128 el)
    unsigned base = (unsigned)(size_t)lds;  // LDS address of the array (low 32 bits of the generic ptr are NOT the LDS offset)
    (void)base;
    unsigned ldsaddr = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) unsigned short*)lds) + addr;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ldsaddr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v & 0xffff);
    out[l * 4 + 1] = (unsigned short)((v >> 16) & 0xffff);
    out[l * 4 + 2] = (unsigned short)((v >> 32) & 0xffff);
    out[l * 4 + 3] = (unsigned short)((v >> 48) & 0xffff);
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
