// Which XCD does workgroup i land on?  768 workgroups x 512 threads with 147 KB of LDS (one per CU), like
// the slab GEMM.  Prints the XCC id of the first workgroups and whether id == i % 8 holds.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
    extern __shared__ char lds[];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) { lds[0] = 1; out[blockIdx.x] = (int)(x & 0xf); }
    // keep the workgroup resident for a while so that the first 256 really occupy 256 CUs
    for (volatile int i = 0; i < 2000; ++i) {}
}
int main() {
    int n = 768, *d, h[768];
    (void)hipMalloc(&d, n * 4);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 147456, 0, d);
    (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < n; ++i) ok += h[i] == i % 8;
    printf("xcc == i %% 8 for %d of %d workgroups\nfirst 48:", ok, n);
    for (int i = 0; i < 48; ++i) printf(" %d", h[i]);
    printf("\nwg 256..287:");
    for (int i = 256; i < 288; ++i) printf(" %d", h[i]);
    printf("\n");
    return 0;
}
