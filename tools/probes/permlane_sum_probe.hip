#include <hip/hip_runtime.h>
__global__ void k(float* p) {
    float x = p[threadIdx.x];
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    p[threadIdx.x] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
int main() {
    float h[64], *d; for (int i = 0; i < 64; ++i) h[i] = (i >> 4) == 0 ? 1 : (i >> 4) == 1 ? 10 : (i >> 4) == 2 ? 100 : 1000;
    for (int i = 0; i < 64; ++i) h[i] += (i & 15) * 0.001f;
    hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 7) printf("%d:%.3f ", i, h[i]); printf("\n");
    return 0;
}
