// One-wave-per-SIMD form of the slab GEMM / implicit-GEMM conv (gemm_mfma.hip) for the MFMA-bound launches: bf16 in, bf16 out,
// plain epilogue (bias [+ ReLU]); 256 x 256 tiles like the slab kernel's tallest form, but FOUR waves (2 x 2), each alone on its
// SIMD with a 128 x 128 patch = 64 accumulator fragments = 256 registers (the accumulation half of the 512-register file a lone
// wave owns).
//
// Why (the step shape of the 8-wave kernel): with two waves per SIMD the issue arbiter serves the older wave first - waves
// 0-3 leave a K step ~1000 cycles before waves 4-7 and idle at the step's barrier; a step takes ~2900 cycles for 2048 of MFMAs.  A
// lone wave per SIMD has nobody to wait for but the barrier itself, and a 128 x 128 patch reads 16 operand fragments per 64 MFMAs
// where a 128 x 64 patch reads 12 per 32: a third less LDS traffic per flop (the K = 768 pointwise launches have the LDS, the operand
// ingest and the MFMA pipe all within 5 % of each other per step).  What it costs: nothing hides a fragment read's latency but
// the wave's own schedule - the reads run two row blocks ahead of their MFMAs in a pinned order.
//
// MEASURED (r05, profiles/r05_v23_quad_ab.txt): bit-identical and 8-13 % SLOWER than the 8-wave kernel (decoder conv1 186-188 vs
// 164-166 us inside the C2 forward; C3 pw1 289 vs 257-261 us) - the K step is ~3600 cycles against ~3190.  What the lone wave pays
// that the 8-wave form hides: its share of the operand DMAs (17 buffer_load ... lds per wave and step, each ~60-100 cycles of the
// wave's issue - in the 8-wave kernel they sit on the four OLDER waves, which otherwise idle at the barrier) and the fragment-read
// latency at the top of each 32-k chunk.  Spreading the DMAs between the MFMAs exposes the last pieces' round trip at the step's
// closing wait (measured in gemm_persist.hip); a third operand stage does not fit (3 x 68 KB).  Off by default (knob 251); kept as
// the checked A/B form of "one wave per SIMD" for this GEMM.
//
// Arithmetic per output element = the slab kernel's: the same LDS images and fragment maps, the same MFMA (v_mfma_f32_16x16x32_bf16,
// weights as the first operand) in the same order (channel block outer, tap inner, two 32-k chunks per step), the same epilogue
// expression.  Bit-identical (tests/test_gpu_ops.py::test_quad_gemm_is_bit_identical_to_the_slab_kernel).
#include <hip/hip_runtime.h>

#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {
namespace {

constexpr int Q_ROWB = 128, Q_BN = 256, Q_BM = 256, Q_KE = 64;
__device__ inline int q_swz(int row, int slot) { return row * Q_ROWB + ((slot ^ (row & 7)) << 4); }  // = gemm_mfma.hip's swz
__device__ inline int q_wcol(int ni, int fgq) { return (ni >> 1) * 32 + fgq * 8 + (ni & 1) * 4; }    // = wcol
__device__ inline int q_wswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }         // = wswz

constexpr int Q_GROUPS = (Q_BM + 30 + 7) / 8;  // 8-row (1 KiB) DMA groups of the slab, k <= 31
constexpr int Q_SI = (Q_GROUPS + 3) / 4;       // slab DMA instructions per wave (4 waves)
constexpr int Q_SLAB_B = Q_SI * 4 * 1024;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_quad_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16;
    __shared__ __attribute__((aligned(16))) unsigned char slab0[Q_SLAB_B];
    __shared__ __attribute__((aligned(16))) unsigned char slab1[Q_SLAB_B];
    __shared__ __attribute__((aligned(16))) unsigned char wt0[Q_BN * Q_ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char wt1[Q_BN * Q_ROWB];
    __shared__ __attribute__((aligned(16))) float sbias[Q_BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, nutt = p.M / S;
    const int tiles_n = (p.N + Q_BN - 1) / Q_BN, tiles_m = (S + Q_BM - 1) / Q_BM;
    int bid = blockIdx.x;
    {   // XCD-contiguous tile order (workgroup i runs on XCD i % 8): the column tiles of a row tile share its slab in one L2
        const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, xcd = bid & 7;
        bid = xcd * per + (xcd < rem ? xcd : rem) + (bid >> 3);
    }
    const int bn = bid % tiles_n;
    bid /= tiles_n;
    const int tm = bid % tiles_m, ub = bid / tiles_m;
    if (ub >= nutt) return;
    const int t0 = tm * Q_BM, n0 = bn * Q_BN;
    const int ntap = p.taps, ncc = p.Cin / Q_KE;
    const T* __restrict__ Xu = (const T*)p.X + (size_t)ub * S * p.ldx;
    constexpr unsigned OOB = 0xFFFFF000u;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)Xu, 0, (unsigned)(((size_t)(S - 1) * p.ldx + p.Cin) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * sizeof(T)), 0x00020000);
    // every wave issues its quarter of a step's DMAs: piece i of wave w = LDS KiB (i * 4 + w) = 8 rows
    unsigned svoff[Q_SI], wvoff[8];
#pragma unroll
    for (int i = 0; i < Q_SI; ++i) {
        const int P = (i * 4 + wave) * 64 + lane, row = P >> 3, ps = P & 7;
        const int t = t0 - p.pad + row;
        const bool ok = (t >= 0) & (t < S) & (row < Q_BM + ntap - 1);
        svoff[i] = ok ? (unsigned)t * (unsigned)(p.ldx * (int)sizeof(T)) + (unsigned)((ps ^ (row & 7)) << 4) : OOB;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int P = (i * 4 + wave) * 64 + lane, row = P >> 3, ps = P & 7;
        const int n = n0 + row;
        wvoff[i] = n < p.N ? (unsigned)n * (unsigned)(p.K * (int)sizeof(T)) + (unsigned)((ps ^ q_wswz(row)) << 4) : OOB;
    }
    auto issue_slab = [&](unsigned char* dst, int cc) {
#pragma unroll
        for (int i = 0; i < Q_SI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(dst + (i * 4 + wave) * 1024), 16, svoff[i],
                                                     cc * Q_ROWB, 0, 0);
    };
    auto issue_w = [&](unsigned char* dst, int cc, int tap) {
        const int koff = (tap * p.Cin + cc * Q_KE) * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + (i * 4 + wave) * 1024), 16, wvoff[i], koff,
                                                     0, 0);
    };

    f32x4_t acc[8][8];  // [ni][mi]
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int xrow0 = wm * 128 + fr;
    int woff[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int wrow = wn * 128 + q_wcol(i, fr >> 2) + (fr & 3);
            woff[i][ks] = wrow * Q_ROWB + (((ks * 4 + fg) ^ q_wswz(wrow)) << 4);
        }
    // A 32-k chunk: the eight weight fragments, then row block by row block one activation fragment and its eight MFMAs; the
    // activation fragments run two blocks ahead (a block's MFMAs are 128 cycles of the pipe).  Accumulator [ni][mi] receives its
    // chunks in the slab kernel's order.
    auto compute = [&](const unsigned char* sl, const unsigned char* wt, int tap) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[8], fx[3];
#pragma unroll
            for (int i = 0; i < 8; ++i) fw[i] = *(const uint4*)(wt + woff[i][ks]);
            fx[0] = *(const uint4*)(sl + q_swz(xrow0 + tap, ks * 4 + fg));
            fx[1] = *(const uint4*)(sl + q_swz(xrow0 + 16 + tap, ks * 4 + fg));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                if (mi + 2 < 8) fx[(mi + 2) % 3] = *(const uint4*)(sl + q_swz(xrow0 + (mi + 2) * 16 + tap, ks * 4 + fg));
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) Mma16<T>::step(fw[ni], fx[mi % 3], acc[ni][mi]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#define Q_STEP(slab_cur, slab_nxt, w_cur, w_nxt, ccv, tapv)                      \
    {                                                                            \
        dma_drain();                                                             \
        __syncthreads();                                                         \
        int ntp = (tapv) + 1, ncb = (ccv);                                       \
        if (ntp == ntap) { ntp = 0; ++ncb; }                                     \
        if (ncb < ncc) issue_w(w_nxt, ncb, ntp);                                 \
        if ((tapv) == 0 && (ccv) + 1 < ncc) issue_slab(slab_nxt, (ccv) + 1);     \
        compute(slab_cur, w_cur, (tapv));                                        \
    }
    issue_slab(slab0, 0);
    issue_w(wt0, 0, 0);
    {
        const int n = n0 + tid;
        sbias[tid] = (p.bias && n < p.N) ? p.bias[n] : 0.f;  // (published by the first step's barrier)
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < ncc; cc += 2) {
        for (int tap = 0; tap < ntap; tap += 2) {
            Q_STEP(slab0, slab1, wt0, wt1, cc, tap)
            if (tap + 1 < ntap) Q_STEP(slab0, slab1, wt1, wt0, cc, tap + 1)
        }
        if (cc + 1 < ncc) {
            for (int tap = 0; tap < ntap; tap += 2) {
                Q_STEP(slab1, slab0, wt1, wt0, cc + 1, tap)
                if (tap + 1 < ntap) Q_STEP(slab1, slab0, wt0, wt1, cc + 1, tap + 1)
            }
        }
    }
#undef Q_STEP
    // ---- epilogue: v = act(acc + bias), 8 consecutive channels per 16-byte store
    T* __restrict__ C = (T*)p.C + (size_t)ub * S * p.ldc;
    const bool fulln = n0 + Q_BN <= p.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 128 + j * 32 + fg * 8, n = n0 + nl;
        float bv[8];
        {
            const float4 b0 = *(const float4*)(sbias + nl), b1 = *(const float4*)(sbias + nl + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
            bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
        if (!fulln && n >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int t = t0 + wm * 128 + mi * 16 + fr;
            if (t >= S) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v[r] = acc[2 * j + (r >> 2)][mi][r & 3] + bv[r];
                if (p.relu) v[r] = fmaxf(v[r], 0.f);
            }
            T* dst = (T*)((char*)C + (unsigned)(t * p.ldc + n) * (unsigned)sizeof(T));
            if (fulln || n + 7 < p.N) {
                *(uint4*)dst = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (n + r < p.N) dst[r] = Num<T>::from_f32(v[r]);
            }
        }
    }
#else
    (void)p;
#endif
}

}  // namespace

// bf16 plain-epilogue launches the slab kernel would run at 256-row tiles
bool gemm_quad_supported(const GemmArgs& a, int in_dtype, int out_dtype) {
    if (in_dtype != FS2_BF16 || out_dtype != FS2_BF16) return false;
    if (a.ln_g || a.dot_w || a.z_out || a.res || a.gate || a.zero_rows || a.C_lo || a.split || a.w_presplit || a.ksplit > 1 || a.drop_p > 0.f) return false;
    if (a.rs_stats || a.stats_out || a.epi_res || a.head_out || !a.C) return false;
    if (!(a.taps & 1) || a.taps > 31 || a.Cin % 64 || a.K != a.taps * a.Cin || a.N < 192 || a.N % 8) return false;
    const int S = a.taps == 1 ? a.M : a.S;
    if (S <= 0 || a.M % S) return false;
    if (a.ldx % 8 || a.ldc % 8) return false;
    if ((size_t)S * a.ldx * 2 >= 0xFFFFF000ull || (size_t)S * a.ldc * 2 >= 0xFFFFF000ull || (size_t)a.N * a.K * 2 >= 0xFFFFF000ull) return false;
    return true;
}

int launch_gemm_quad(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.taps == 1) a.S = a.M;
    a.pad = (a.taps - 1) / 2;
    const int tiles = (a.M / a.S) * ((a.S + Q_BM - 1) / Q_BM) * ((a.N + Q_BN - 1) / Q_BN);
    hipLaunchKernelGGL(gemm_quad_kernel, dim3(tiles), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
