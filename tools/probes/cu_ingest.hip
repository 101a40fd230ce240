// Probe: how many bytes per second can ONE CU pull out of L2 into LDS, by path?  256 workgroups (one per CU, 512 threads) each walk
// the same PANEL bytes (L2 / MALL resident, as a GEMM's weight panel is) in 32-KiB steps, double-buffered with one barrier per
// step, the slab kernel's scheme.
//   MODE 0: buffer_load_dwordx4 ... lds (LDS-DMA), issued by all 8 waves (4 x 1 KiB each per step)
//   MODE 1: LDS-DMA issued by 4 of the 8 waves (8 x 1 KiB each)
//   MODE 2: global_load_dwordx4 -> VGPR -> ds_write_b128 (4 x 16 B per thread per step), loads of step s+1 issued before the
//           barrier of step s, written after it
//   MODE 3: as 2, two steps of loads in flight
//   MODE 4 / 5: LDS-DMA by all 8 waves into a ring of 3 / 4 stages (2 / 3 steps in flight), counted vmcnt, one barrier per step
//   MODE 6: global_load_dwordx4 -> VGPR only (no LDS stage: the single-launch predictor's weight path), a 4-deep register ring,
//           every thread 64 B per step, consumed by adds; no barrier
//   MODE 7: BOTH at once - the 32 KiB of a step by LDS-DMA (mode 0) and another 32 KiB by the register path (mode 6's ring): does a CU
//           ingest more through two paths than through one?  (reported: the SUM of both streams)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_ingest tools/probes/cu_ingest.hip && /tmp/cu_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int STEP = 32768;
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* w, int panel, int reps, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char b0[STEP];
    __shared__ __attribute__((aligned(16))) unsigned char b1[STEP];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)panel, 0x00020000);
    const int nsteps = panel / STEP;
    float acc = 0.f;
    auto dma = [&](unsigned char* dst, int s) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (i * 8 + wave) * 1024), 16,
                                                         (unsigned)(s * STEP + (i * 8 + wave) * 1024 + lane * 16), 0, 0, 0);
        } else if (wave < 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (i * 4 + wave) * 1024), 16,
                                                         (unsigned)(s * STEP + (i * 4 + wave) * 1024 + lane * 16), 0, 0, 0);
        }
    };
    auto consume = [&](const unsigned char* src) {  // every thread reads 64 B of the step back (a GEMM wave reads far more)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = *(const float4*)(src + ((i * 512 + tid) * 16));
            acc += v.x + v.w;
        }
    };
    for (int r = 0; r < reps; ++r) {
        if (MODE == 6 || MODE == 7) {
            uint4 q[4][4];
            auto ld = [&](uint4 (&x)[4], int s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = *(const uint4*)(w + (MODE == 7 ? ((size_t)panel + (size_t)s * STEP) % (size_t)(8 << 20) : (size_t)s * STEP) + (i * 512 + tid) * 16);
            };
            auto use = [&](const uint4 (&x)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += __uint_as_float(x[i].x) + __uint_as_float(x[i].w);
            };
            // (mode 7 reads its register stream from the bytes BEHIND the panel - another L2-resident region - so that the two streams
            // do not hit the same lines)
            ld(q[0], 0); if (nsteps > 1) ld(q[1], 1); if (nsteps > 2) ld(q[2], 2);
            if (MODE == 7) dma(b0, 0);
#pragma unroll 4
            for (int s = 0; s < nsteps; ++s) {
                if (MODE == 7) {
                    // the DMA of step s must have landed; the register loads of steps s+1.. (4 per step, issued after it) may fly
                    if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (s + 1 < nsteps) dma((s & 1) ? b0 : b1, s + 1);
                }
                if (s + 3 < nsteps) ld(q[(s + 3) & 3], s + 3);
                use(q[s & 3]);
                if (MODE == 7) consume((s & 1) ? b1 : b0);
            }
            if (MODE == 7) __syncthreads();
        } else if (MODE >= 4) {
            constexpr int NST = MODE - 1;  // 3 or 4 stages
            __shared__ __attribute__((aligned(16))) unsigned char ring[4][STEP];
            unsigned char* const st0 = &ring[0][0];
            auto dmar = [&](int slot, int s) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st0 + slot * STEP + (i * 8 + wave) * 1024), 16,
                                                             (unsigned)(s * STEP + (i * 8 + wave) * 1024 + lane * 16), 0, 0, 0);
            };
            for (int s = 0; s < NST - 1 && s < nsteps; ++s) dmar(s, s);
            int slot = 0;
            for (int s = 0; s < nsteps; ++s) {
                // step s landed = all but the (NST - 2) younger steps' DMAs (4 instructions per step per wave) are done
                if (s + NST - 2 < nsteps) {
                    if (NST == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (s + NST - 1 < nsteps) dmar((slot + NST - 1) % NST, s + NST - 1);
                consume(st0 + slot * STEP);
                slot = (slot + 1) % NST;
            }
            __syncthreads();
        } else if (MODE <= 1) {
            dma(b0, 0);
            for (int s = 0; s < nsteps; ++s) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (s + 1 < nsteps) dma((s & 1) ? b0 : b1, s + 1);
                consume((s & 1) ? b1 : b0);
            }
            __syncthreads();
        } else {
            uint4 q[2][4];
            auto ld = [&](uint4 (&x)[4], int s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = *(const uint4*)(w + (size_t)s * STEP + (i * 512 + tid) * 16);
            };
            auto st = [&](unsigned char* dst, uint4 (&x)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint4*)(dst + (i * 512 + tid) * 16) = x[i];
            };
            ld(q[0], 0);
            if (MODE == 3 && nsteps > 1) ld(q[1], 1);
            for (int s = 0; s < nsteps; ++s) {
                if (MODE == 2) {
                    st((s & 1) ? b1 : b0, q[0]);
                    if (s + 1 < nsteps) ld(q[0], s + 1);
                } else {
                    if (s & 1) { st(b1, q[1]); if (s + 2 < nsteps) ld(q[1], s + 2); }
                    else { st(b0, q[0]); if (s + 2 < nsteps) ld(q[0], s + 2); }
                }
                __syncthreads();
                consume((s & 1) ? b1 : b0);
                __syncthreads();
            }
        }
    }
    out[blockIdx.x * 512 + tid] = acc;
}
template <int MODE> void run(const unsigned char* w, float* o, int panel, int grid) {
    const int reps = (64 << 20) / panel;
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, w, panel, 2, o);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, w, panel, reps, o);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)panel * reps / (ms * 1e-3) / 1e9;
    const double mult = MODE == 7 ? 2.0 : 1.0;  // two 32-KiB streams per step
    printf("mode %d  panel %5d KiB  %3d workgroups: %6.1f GB/s per CU, %5.2f TB/s chip\n", MODE, panel >> 10, grid, per_cu * mult, per_cu * mult * grid / 1e3);
}
int main() {
    unsigned char* w; float* o;
    (void)hipMalloc(&w, 8 << 20); (void)hipMemset(w, 1, 8 << 20); (void)hipMalloc(&o, 256 * 512 * 4);
    for (int panel : {512 << 10, 4 << 20}) {
        run<0>(w, o, panel, 256); run<1>(w, o, panel, 256); run<2>(w, o, panel, 256); run<3>(w, o, panel, 256);
        run<4>(w, o, panel, 256); run<5>(w, o, panel, 256); run<6>(w, o, panel, 256); run<7>(w, o, panel, 256);
    }
    run<0>(w, o, 512 << 10, 32); run<2>(w, o, 512 << 10, 32);
    return 0;
}
