// Which hardware registers tell the two co-resident workgroups of a CU apart?  480 workgroups x 256 threads x 62 KB of LDS (the
// predictor's launch shape): HW_ID (id 4) and LDS_ALLOC (id 6) of wave 0 of every workgroup, with XCC_ID and a start stamp.
// hipcc --offload-arch=gfx950 -O2 tools/probes/hwid_probe.hip -o tools/probes/bin/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out) {
    __shared__ unsigned char pad[61952];
    pad[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        unsigned la = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);
        unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        unsigned long long t = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = la; out[blockIdx.x * 4 + 2] = xcc + pad[5] - 1; out[blockIdx.x * 4 + 3] = (unsigned)t;
    }
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);  // stay resident so that the second round cannot reuse a slot
}
int main() {
    const int n = 480;
    unsigned* d; hipMalloc(&d, n * 16);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        if (i < 24 || (i >= 252 && i < 276) || i >= 470)
            printf("wg %3d xcc %u hw_id %08x (wave %u simd %u cu %u sh %u se %u tg %u) lds_alloc %08x (base %u) t %u\n", i, h[i * 4 + 2], h[i * 4],
                   h[i * 4] & 15, (h[i * 4] >> 4) & 3, (h[i * 4] >> 8) & 15, (h[i * 4] >> 12) & 1, (h[i * 4] >> 13) & 7, (h[i * 4] >> 16) & 15,
                   h[i * 4 + 1], h[i * 4 + 1] & 0xff, h[i * 4 + 3]);
    return 0;
}
