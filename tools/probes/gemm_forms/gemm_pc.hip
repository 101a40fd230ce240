// Producer / consumer form of the pointwise slab GEMM (gemm_mfma.hip), the step gemm_quad.hip / gemm_ring.hip pointed at: bf16 in,
// bf16 out, taps == 1, plain epilogue (bias [+ ReLU]).  192 x 256 tiles on EIGHT waves of two kinds:
//   * waves 0-3 (the older wave of each SIMD, 2 x 2) only multiply: a 96 x 128 accumulator patch each (48 fragments = 192
//     registers), fragments from LDS, never a vector-memory instruction - the r05 probes put ~100 issue cycles on every
//     buffer_load ... lds for the wave that issues it, which a wave with a back-to-back MFMA stream cannot spare;
//   * waves 4-7 only request: each K step (32 wide) they issue the pieces of the stage FOUR steps ahead into a 5-stage ring, wait
//     (counted) until the stage two steps ahead has landed, and meet the consumers at the step's one barrier.  Their instructions
//     are scalar + vector-memory: they issue beside the older wave's MFMAs instead of competing with them.
// After barrier q the consumers may read stages <= q + 2 and the producers may overwrite the buffer of stage q; a consumer reads
// its next step's weight fragments INTO the registers of the current ones, one after each MFMA of the step's last row block.
// MEASURED (r05, profiles/r05_v27_pc_ab.txt, r05_v28_pc_probe.txt): bit-identical on the first run and 5-13 % SLOWER than the slab
// kernel (C3 in-projection 230-233 vs 206-212 us, pw1 295 vs 261, conv2 217-222 vs 208; the persistent form: 194 / 242 / 199).  The
// probes: no requests inside the K loop 155 us (conv2: 1.49 PF - the consumers' MFMA stream at the power-limited clock), requests
// WITHOUT the counted wait 222 (= shipped: it is not latency), requests that are all out of range - the same instructions, zero
// fills, no L2 traffic - 143.  So the request INSTRUCTIONS are free here (they were ~100 cycles each for a lone MFMA wave), and
// what costs 40 % is the operands actually arriving: 28 KB per 32-k step through the CU's L2 -> LDS path beside 56 KB of
// fragment reads.  All four forms of this GEMM (8-wave slab, persistent, lone wave + ring, producer / consumer) land at 0.72-0.77 us
// per 32-k step of a 192-row pointwise tile: the bound of the K = 768 launches is the operand path into the CU, not how the waves
// are arranged around it.  Off by default (knob 253).
//
// Same 32-k chunks in the same order per output element as the slab kernel: bit-identical
// (tests/test_gpu_ops.py::test_pc_gemm_is_bit_identical_to_the_slab_kernel).
#include <hip/hip_runtime.h>

#include "fs2_common.h"
#include "fs2_kernels.h"

#ifndef PC_PROBE
#define PC_PROBE 0  // timing probes (wrong results): 1 = no requests inside the K loop, 2 = requests without the counted wait, 3 = out-of-range requests (zero fills: no L2 traffic)
#endif
namespace fs2 {
namespace {

constexpr int PC_BM = 192, PC_BN = 256, PC_NS = 5;
constexpr int PC_WST = 256 * 64, PC_XST = 192 * 64;
constexpr int PC_OFF_X = PC_NS * PC_WST, PC_LDS = PC_OFF_X + PC_NS * PC_XST;
constexpr int PC_NP = 7;  // pieces per producer wave and stage: 4 of the weight tile + 3 of the activation tile
__device__ inline int pc_xslot(int L, int row) { return ((L & 1) << 1) | (((L >> 1) ^ (row >> 2)) & 1); }  // SlabSwizzle, 4 slots
__device__ inline int pc_wslot(int L, int row) { return L ^ ((row >> 2) & 3); }

typedef __attribute__((ext_vector_type(4))) int pc_rsrc_t;
__device__ inline pc_rsrc_t pc_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long u = (unsigned long long)(uintptr_t)base;
    pc_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(u >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void pc_dma(unsigned m0v, unsigned voff, const pc_rsrc_t& rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
#define PC_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define PC_BARRIER()                       \
    {                                      \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    }

__global__ __launch_bounds__(512) void gemm_pc_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[PC_LDS];
    __shared__ __attribute__((aligned(16))) float sbias[PC_BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, nutt = p.M / S;
    const int tiles_n = (p.N + PC_BN - 1) / PC_BN, tiles_m = (S + PC_BM - 1) / PC_BM;
    int bid = blockIdx.x;
    {   // XCD-contiguous tile order (workgroup i runs on XCD i % 8)
        const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, xcd = bid & 7;
        bid = xcd * per + (xcd < rem ? xcd : rem) + (bid >> 3);
    }
    const int bn = bid % tiles_n;
    bid /= tiles_n;
    const int tm = bid % tiles_m, ub = bid / tiles_m;
    if (ub >= nutt) return;
    const int t0 = tm * PC_BM, n0 = bn * PC_BN;
    const int nst = p.Cin / 32;
    if (tid < PC_BN) {
        const int n = n0 + tid;
        sbias[tid] = (p.bias && n < p.N) ? p.bias[n] : 0.f;  // (published by the prologue's barrier)
    }

    if (wave >= 4) {
        // ---------------- producers ----------------
        const int pw = wave - 4;
        const T* Xu = (const T*)p.X + (size_t)ub * S * p.ldx;
        constexpr unsigned OOB = 0xFFFFF000u;
        const pc_rsrc_t xrs = pc_make_rsrc(Xu, (unsigned)(((size_t)(S - 1) * p.ldx + p.Cin) * sizeof(T)));
        const pc_rsrc_t wrs = pc_make_rsrc(p.W, (unsigned)((size_t)p.N * p.K * sizeof(T)));
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
        // a piece = 1 KiB = 16 LDS rows x 4 slots, written lane-linearly: lane l -> row l >> 2, physical slot l & 3 (the swizzle is
        // applied on the global side)
        unsigned wvo[4], xvo[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (j * 4 + pw) * 16 + (lane >> 2), ps = lane & 3;
            const int n = n0 + row, L = ps ^ ((row >> 2) & 3);
            wvo[j] = n < p.N ? (unsigned)n * (unsigned)(p.K * 2) + (unsigned)(L << 4) : OOB;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int row = (j * 4 + pw) * 16 + (lane >> 2), ps = lane & 3;
            const int t = t0 + row;
            const int L = (ps >> 1) | ((((ps & 1) ^ (row >> 2)) & 1) << 1);  // inverse of pc_xslot
            xvo[j] = t < S ? (unsigned)t * (unsigned)(p.ldx * 2) + (unsigned)(L << 4) : OOB;
        }
        auto issue_stage = [&](int st) {
            const int slot = st % PC_NS;
            const bool live = PC_PROBE == 3 ? st < 4 : st < nst;
            const unsigned soff = (unsigned)(st * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) pc_dma(lds0 + (unsigned)(slot * PC_WST + (j * 4 + pw) * 1024), live ? wvo[j] : OOB, wrs, soff);
#pragma unroll
            for (int j = 0; j < 3; ++j) pc_dma(lds0 + (unsigned)(PC_OFF_X + slot * PC_XST + (j * 4 + pw) * 1024), live ? xvo[j] : OOB, xrs, soff);
        };
        issue_stage(0);
        issue_stage(1);
        issue_stage(2);
        issue_stage(3);
        PC_VMCNT(2 * PC_NP);  // stages 0 and 1
        PC_BARRIER();
        for (int q = 0; q < nst; ++q) {
            if (PC_PROBE != 1) issue_stage(q + 4);
            if (PC_PROBE == 0) PC_VMCNT(2 * PC_NP);  // stage q + 2 has landed (q + 3 and q + 4 may be on their way)
            PC_BARRIER();
        }
        PC_VMCNT(0);
        return;
    }

    // ---------------- consumers ----------------
    f32x4_t acc[8][6];  // [ni][mi]
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int xrow0 = wm * 96 + fr;
    unsigned woffE, woffO, xoff;
    {
        const int rE = wn * 128 + (fr >> 2) * 8 + (fr & 3), rO = rE + 4;
        woffE = (unsigned)(rE * 64 + (pc_wslot(fg, rE) << 4));
        woffO = (unsigned)(rO * 64 + (pc_wslot(fg, rO) << 4));
        xoff = (unsigned)(PC_OFF_X + xrow0 * 64 + (pc_xslot(fg, xrow0) << 4));  // row block mi: + mi KiB (16 rows keep the swizzle term)
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    PC_BARRIER();
    uint4 fw[8], fx[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *(const uint4*)(lds + ((i & 1) ? woffO : woffE) + (i >> 1) * 2048);
    fx[0] = *(const uint4*)(lds + xoff);
    fx[1] = *(const uint4*)(lds + xoff + 1024);
    int slot = 0;
    for (int q = 0; q < nst; ++q) {
        const int nslot = slot + 1 == PC_NS ? 0 : slot + 1;
        const unsigned xa = xoff + (unsigned)(slot * PC_XST), xn = xoff + (unsigned)(nslot * PC_XST);
        const unsigned wn0 = (unsigned)(nslot * PC_WST);
#pragma unroll
        for (int mi = 0; mi < 6; ++mi) {
            // the activation fragment two row blocks ahead ("blocks 6 and 7" = the next step's 0 and 1, which is stage q + 1:
            // visible since the previous barrier)
            if (mi + 2 < 6) fx[(mi + 2) % 3] = *(const uint4*)(lds + xa + (mi + 2) * 1024);
            else fx[(mi + 2) % 3] = *(const uint4*)(lds + xn + (mi + 2 - 6) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            if (mi < 5) {
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) Mma16<T>::step(fw[ni], fx[mi % 3], acc[ni][mi]);
            } else {
                // the last row block: behind each MFMA its weight fragment is replaced by the next step's
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) {
                    Mma16<T>::step(fw[ni], fx[mi % 3], acc[ni][mi]);
                    fw[ni] = *(const uint4*)(lds + wn0 + ((ni & 1) ? woffO : woffE) + (ni >> 1) * 2048);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PC_BARRIER();
        slot = nslot;
    }

    // ---- epilogue: v = act(acc + bias), 8 consecutive channels per 16-byte store
    T* __restrict__ C = (T*)p.C + (size_t)ub * S * p.ldc;
    const bool fulln = n0 + PC_BN <= p.N;
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
        for (int mi = 0; mi < 6; ++mi) {
            const int t = t0 + wm * 96 + mi * 16 + fr;
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

bool gemm_pc_supported(const GemmArgs& a, int in_dtype, int out_dtype) {
    if (in_dtype != FS2_BF16 || out_dtype != FS2_BF16) return false;
    if (a.ln_g || a.dot_w || a.z_out || a.res || a.gate || a.zero_rows || a.C_lo || a.split || a.w_presplit || a.ksplit > 1 || a.drop_p > 0.f) return false;
    if (a.rs_stats || a.stats_out || a.epi_res || a.head_out || !a.C) return false;
    if (a.taps != 1 || a.Cin % 64 || a.Cin < 128 || a.K != a.Cin || a.N < 192 || a.N % 8) return false;
    if (a.ldx % 8 || a.ldc % 8) return false;
    if ((size_t)a.M * a.ldx * 2 >= 0xFFFFF000ull || (size_t)a.M * a.ldc * 2 >= 0xFFFFF000ull || (size_t)a.N * a.K * 2 >= 0xFFFFF000ull) return false;
    return true;
}

int launch_gemm_pc(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    a.S = a.M;
    const int tiles = ((a.S + PC_BM - 1) / PC_BM) * ((a.N + PC_BN - 1) / PC_BN);
    hipLaunchKernelGGL(gemm_pc_kernel, dim3(tiles), dim3(512), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
