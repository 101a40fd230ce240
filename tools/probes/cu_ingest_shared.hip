// Probe: does a CU's register-path weight stream (global_load_dwordx4 -> VGPR, a 4-deep register ring, no LDS stage - the weight
// path of vocoder_resblock / predictor_fused) get faster when several waves of the workgroup request the SAME lines (vector-L1 hits)?
// 256 workgroups (one per CU) x 8 waves; every wave requests 4 KiB per step (4 x dwordx4 per lane); groups of SH waves request the
// same 4 KiB.  SH = 1: 32 KiB of distinct lines per workgroup-step (cu_ingest.hip mode 6), SH = 8: one 4 KiB for everybody.
// Reported: bytes DELIVERED into registers per CU (8 x 4 KiB per step) and the distinct bytes behind them.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_ingest_shared tools/probes/cu_ingest_shared.hip && /tmp/cu_ingest_shared
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SH, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k(const unsigned char* w, int panel, int reps, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int STEP = (NW / SH) * 4096;  // distinct bytes per workgroup-step
    const int nsteps = panel / STEP;
    const unsigned char* base = w + (wave / SH) * 4096 + lane * 16;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        uint4 q[4][4];
        auto ld = [&](uint4 (&x)[4], int s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = *(const uint4*)(base + (size_t)s * STEP + i * 1024);
        };
        auto use = [&](const uint4 (&x)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += __uint_as_float(x[i].x) + __uint_as_float(x[i].w);
        };
        ld(q[0], 0); ld(q[1], 1); ld(q[2], 2);
#pragma unroll 4
        for (int s = 0; s < nsteps; ++s) {
            if (s + 3 < nsteps) ld(q[(s + 3) & 3], s + 3);
            use(q[s & 3]);
        }
    }
    out[blockIdx.x * NW * 64 + tid] = acc;
}
template <int SH, int NW> void run(const unsigned char* w, float* o, int panel) {
    constexpr int STEP = (NW / SH) * 4096;
    const int reps = (int)((64ll << 20) * (NW / SH) / 8 / panel) > 0 ? (int)((64ll << 20) * (NW / SH) / 8 / panel) : 1;
    hipLaunchKernelGGL((k<SH, NW>), dim3(256), dim3(NW * 64), 0, 0, w, panel, 2, o);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SH, NW>), dim3(256), dim3(NW * 64), 0, 0, w, panel, reps, o);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)(panel / STEP) * reps;
    const double delivered = steps * NW * 4096 / (ms * 1e-3) / 1e9, distinct = steps * STEP / (ms * 1e-3) / 1e9;
    printf("waves %d  share %d  panel %4d KiB: delivered %6.1f GB/s per CU (%5.1f B/clk at 2.4 GHz), distinct %6.1f GB/s per CU\n", NW, SH, panel >> 10,
           delivered, delivered / 2.4, distinct);
}
int main() {
    unsigned char* w; float* o;
    (void)hipMalloc(&w, 8 << 20); (void)hipMemset(w, 1, 8 << 20); (void)hipMalloc(&o, 256 * 512 * 4);
    for (int panel : {512 << 10, 2 << 20}) {
        run<1, 8>(w, o, panel); run<2, 8>(w, o, panel); run<4, 8>(w, o, panel); run<8, 8>(w, o, panel);
        run<1, 4>(w, o, panel); run<2, 4>(w, o, panel); run<4, 4>(w, o, panel);
    }
    return 0;
}
