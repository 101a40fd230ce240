// One wave per SIMD, operands requested THREE K-steps ahead: the form of the slab GEMM / implicit-GEMM conv (gemm_mfma.hip) that
// gemm_quad.hip's measurement asked for.  bf16 in, bf16 out, plain epilogue (bias [+ ReLU]); 256 x 256 tiles on FOUR waves (2 x 2),
// each alone on its SIMD with a 128 x 128 accumulator patch (256 registers of the 512 a lone wave owns).
//
// What gemm_quad.hip showed (profiles/r05_v23_quad_ab.txt, r05_v24_nodma_probe.txt): a lone wave per SIMD runs a K step's 128 MFMAs
// with none of the 8-wave kernel's older-wave / younger-wave skew - with the operand DMAs taken out it is 20-26 % faster than the
// shipped kernel (decoder conv1 164 vs 205 us back to back, C3 pw1 210 vs 276) - but it has to issue its share of the DMAs itself,
// and 17 buffer_load ... lds in a bunch block it for ~1000 cycles (the CU's texture path takes a 1-KiB piece per ~16-32 cycles).  Here
// the K step is 32 wide (64 MFMAs per wave, eight row blocks of eight), the operands sit in a 4-stage ring, and the pieces of stage
// q + 3 are requested ONE PER ROW BLOCK inside step q - the texture path never backs up, the pieces have two and a half steps to
// land, and the wave never waits for more than its own fragment reads:
//   * one raw s_barrier per step, between row blocks 5 and 6, behind a COUNTED wait (vmcnt = the pieces of the last two steps):
//     it publishes stage q + 1 (whose first fragments are read in blocks 6 / 7, under the step's last MFMAs) and frees the buffer
//     of stage q - 1 for the requests of step q + 1;
//   * the DMAs are asm statements (hipcc's LDS-DMA scoreboard cannot tell ring slots apart and would drain vmcnt in front of
//     every fragment read; attention_pipe.hip's way), the fragment reads are ordinary loads under sched_barrier fences - hipcc counts
//     lgkmcnt for them itself;
//   * 64-byte LDS rows: activation rows take SlabSwizzle's 4-slot map (conflict-free for sixteen consecutive rows from any start -
//     every conv tap), weight rows slot ^ ((row >> 2) & 3) (conflict-free for the fragment's 4-rows-every-8 pattern); both checked
//     against the bank model of tools/probes/lds_swizzle_sim.py.
// MEASURED (r05, profiles/r05_v25_ring_ab.txt, r05_v26_ring_probe.txt): bit-identical on the first run, and SLOWER than both the 8-wave
// kernel and gemm_quad.hip: decoder conv1 254-259 us back to back (8-wave 197-206, quad 214-222), C3 pw1 285-300 (259, 289-316).
// The probes say why: with no requests inside the K loop the launch takes 179 us (the power-limited MFMA rate: ~1.3-1.4 PF, which
// is also what the 8-wave kernel reaches without its DMAs: 181 us), with the requests but WITHOUT the counted wait 251 us - so it is
// not latency, not the ring and not the barrier: every buffer_load ... lds costs the wave that issues it ~100 cycles in which it
// issues nothing else (5 pieces per 32-k step = ~490 cycles per 1024 of MFMAs; 8 at pointwise launches), wherever in the step it
// sits.  A lone wave per SIMD cannot afford to issue its own operand requests; the 8-wave kernel parks them on the four older
// waves, which have the slack.  What is left to try is a producer / consumer split (four MFMA waves that never touch vector memory
// + four request waves, 192-row tiles so that accumulators + fragments fit 256 registers).  Off by default (knob 252).
//
// Arithmetic per output element = the slab kernel's: the same MFMA (v_mfma_f32_16x16x32_bf16, weights as the first operand) over
// the same 32-k chunks in the same order (64-channel block outer, tap, the block's two halves inner) and the same epilogue
// expression: bit-identical (tests/test_gpu_ops.py::test_ring_gemm_is_bit_identical_to_the_slab_kernel).
#include <hip/hip_runtime.h>

#include "fs2_common.h"
#include "fs2_kernels.h"

#ifndef RG_PROBE
#define RG_PROBE 0  // timing probes (wrong results): 1 = no operand requests inside the K loop, 2 = requests but no counted wait
#endif
#ifndef RG_SINGLE
#define RG_SINGLE 0
#endif
namespace fs2 {
namespace {

constexpr int RG_BM = 256, RG_BN = 256;
constexpr int RG_WST = 256 * 64;      // one weight stage: 256 rows x 32 k
constexpr int RG_XPW = 256 * 64;      // pointwise: one activation stage
constexpr int RG_XCV = 20 * 1024;     // conv: one half-slab (<= 286 rows x 32 channels), 20 pieces
__device__ inline int rg_xslot(int L, int row) { return ((L & 1) << 1) | (((L >> 1) ^ (row >> 2)) & 1); }  // SlabSwizzle, 4 slots
__device__ inline int rg_wslot(int L, int row) { return L ^ ((row >> 2) & 3); }

typedef __attribute__((ext_vector_type(4))) int rg_rsrc_t;
__device__ inline rg_rsrc_t rg_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long u = (unsigned long long)(uintptr_t)base;
    rg_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(u >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void rg_dma(unsigned m0v, unsigned voff, const rg_rsrc_t& rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
#define RG_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// PW: pointwise (taps == 1): a stage = weight tile + activation tile of one 32-k chunk, SPS = 4 activation pieces per wave and step.
// Conv: the half-slabs (64-channel block cb, half h) live in their own 4-ring - both halves of the current block and both of the
// next one; the ten pieces per wave of block cb + 1 are requested SPS per step over the first steps of block cb.
template <bool PW, int SPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_ring_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16;
    constexpr int XST = PW ? RG_XPW : RG_XCV;
    constexpr int OFF_X = 4 * RG_WST, OFF_DUMMY = OFF_X + 4 * XST, OFF_END = OFF_DUMMY + (PW ? 0 : 4096);
    constexpr int P = 4 + SPS;                      // DMA pieces per wave and step
    constexpr int PB = P < 6 ? P : 6;               // ... of which in front of the step's barrier (one per row block 0 .. 5)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[OFF_END];
    __shared__ __attribute__((aligned(16))) float sbias[RG_BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, nutt = p.M / S;
    const int tiles_n = (p.N + RG_BN - 1) / RG_BN, tiles_m = (S + RG_BM - 1) / RG_BM;
    int bid = blockIdx.x;
    {   // XCD-contiguous tile order (workgroup i runs on XCD i % 8)
        const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, xcd = bid & 7;
        bid = xcd * per + (xcd < rem ? xcd : rem) + (bid >> 3);
    }
    const int bn = bid % tiles_n;
    bid /= tiles_n;
    const int tm = bid % tiles_m, ub = bid / tiles_m;
    if (ub >= nutt) return;
    const int t0 = tm * RG_BM, n0 = bn * RG_BN;
    const int ntap = PW ? 1 : p.taps, ncb = p.Cin / 64;  // 64-channel blocks
    const int nst = ncb * ntap * 2;                      // steps: (block, tap, half)
    const T* Xu = (const T*)p.X + (size_t)ub * S * p.ldx;
    constexpr unsigned OOB = 0xFFFFF000u;
    const rg_rsrc_t xrs = rg_make_rsrc(Xu, (unsigned)(((size_t)(S - 1) * p.ldx + p.Cin) * sizeof(T)));
    const rg_rsrc_t wrs = rg_make_rsrc(p.W, (unsigned)((size_t)p.N * p.K * sizeof(T)));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // ---- per-lane DMA offsets.  A piece = 1 KiB = 16 LDS rows x 4 slots, written lane-linearly: lane l -> row l >> 2, physical
    // slot l & 3, which holds the logical slot the swizzle maps there (applied on the global side)
    unsigned wvo[4], xvo[PW ? 4 : 5];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 4 + wave) * 16 + (lane >> 2), ps = lane & 3;
        const int n = n0 + row, L = ps ^ ((row >> 2) & 3);
        wvo[j] = n < p.N ? (unsigned)n * (unsigned)(p.K * 2) + (unsigned)(L << 4) : OOB;
    }
#pragma unroll
    for (int j = 0; j < (PW ? 4 : 5); ++j) {
        const int row = (j * 4 + wave) * 16 + (lane >> 2), ps = lane & 3;
        const int t = t0 - (PW ? 0 : p.pad) + row;
        const int L = (ps >> 1) | ((((ps & 1) ^ (row >> 2)) & 1) << 1);  // inverse of rg_xslot
        const bool ok = (t >= 0) & (t < S) & (row < RG_BM + ntap - 1);
        xvo[j] = ok ? (unsigned)t * (unsigned)(p.ldx * 2) + (unsigned)(L << 4) : OOB;
    }
    // piece u (0 .. P - 1) of the requests made inside step q: the operands of step q + 3 (weights: u < 4), and the activations:
    // pointwise - of step q + 3 too; conv - piece (step within block) * SPS + (u - 4) of the ten of the NEXT block's half-slabs
    int cb = 0, tap = 0, hf = 0;  // of step q
    auto issue_piece = [&](int q, int u) {
        if (u < 4) {
            const int q3 = q + 3;
            // (block, tap, half) of step q + 3
            int h3 = hf + 3, t3 = tap + (h3 >> 1), c3 = cb;
            h3 &= 1;
            while (t3 >= ntap) { t3 -= ntap; ++c3; }
            const unsigned soff = (unsigned)((t3 * p.Cin + c3 * 64 + h3 * 32) * 2);
            const unsigned v = q3 < nst ? wvo[u] : OOB;
            rg_dma(lds0 + (unsigned)((q3 & 3) * RG_WST + (u * 4 + wave) * 1024), v, wrs, soff);
        } else if constexpr (PW) {
            const int q3 = q + 3, j = u - 4;
            const unsigned v = q3 < nst ? xvo[j] : OOB;
            rg_dma(lds0 + (unsigned)(OFF_X + (q3 & 3) * XST + (j * 4 + wave) * 1024), v, xrs, (unsigned)(q3 * 64));
        } else {
            const int sb = tap * 2 + hf;              // step within the block
            const int k = sb * SPS + (u - 4);         // 0 .. 9: half-slab k / 5, piece k % 5 of block cb + 1
            if (k < 10) {
                const int h = k / 5, j = k - h * 5;
                const unsigned v = cb + 1 < ncb ? xvo[j] : OOB;
                rg_dma(lds0 + (unsigned)(OFF_X + (((cb + 1) * 2 + h) & 3) * XST + (j * 4 + wave) * 1024), v, xrs, (unsigned)(((cb + 1) * 64 + h * 32) * 2));
            } else {
                rg_dma(lds0 + (unsigned)(OFF_DUMMY + wave * 1024), OOB, xrs, 0u);  // keeps the count per step constant (a zero fill of 1 KiB nobody reads)
            }
        }
    };

    f32x4_t acc[8][8];  // [ni][mi]
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int xrow0 = wm * 128 + fr;
    // weight fragment ni: LDS row wn * 128 + (ni >> 1) * 32 + (fr >> 2) * 8 + (ni & 1) * 4 + (fr & 3); the swizzle term does not
    // depend on ni >> 1, so two per-lane offsets (ni even / odd) + a multiple of 2 KiB
    unsigned woffE, woffO;
    {
        const int rE = wn * 128 + (fr >> 2) * 8 + (fr & 3), rO = rE + 4;
        woffE = (unsigned)(rE * 64 + (rg_wslot(fg, rE) << 4));
        woffO = (unsigned)(rO * 64 + (rg_wslot(fg, rO) << 4));
    }
    auto xoff_of = [&](int tp) -> unsigned {  // activation fragment of row block 0 at tap tp; row block mi: + mi KiB
        const int row = xrow0 + tp;
        return (unsigned)(row * 64 + (rg_xslot(fg, row) << 4));
    };
    auto read_w = [&](uint4 (&fw)[8], unsigned stage_off, int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
            fw[i] = *(const uint4*)(lds + stage_off + ((i & 1) ? woffO : woffE) + (i >> 1) * 2048);
    };
    auto xstage_of = [&](int q, int c, int h) -> unsigned { return (unsigned)(OFF_X + (PW ? (q & 3) : ((c * 2 + h) & 3)) * XST); };

#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0, 1, 2 (conv: + the half-slabs of blocks 0 and 1)
    float bias_v;
    {
        const int n = n0 + tid;
        bias_v = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int q0 = 0; q0 < 3; ++q0) {
        int t3 = q0 >> 1, c3 = 0;
        const int h3 = q0 & 1;
        while (t3 >= ntap) { t3 -= ntap; ++c3; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            rg_dma(lds0 + (unsigned)(q0 * RG_WST + (u * 4 + wave) * 1024), q0 < nst ? wvo[u] : OOB, wrs, (unsigned)((t3 * p.Cin + c3 * 64 + h3 * 32) * 2));
        if constexpr (PW) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rg_dma(lds0 + (unsigned)(OFF_X + q0 * XST + (j * 4 + wave) * 1024), q0 < nst ? xvo[j] : OOB, xrs, (unsigned)(q0 * 64));
        }
    }
    if constexpr (!PW) {
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const int c = k / 10, h = (k / 5) & 1, j = k % 5;
            rg_dma(lds0 + (unsigned)(OFF_X + ((c * 2 + h) & 3) * XST + (j * 4 + wave) * 1024), c < ncb ? xvo[j] : OOB, xrs, (unsigned)((c * 64 + h * 32) * 2));
        }
    }
    if constexpr (PW) RG_VMCNT(2 * 8); else RG_VMCNT(8);  // stage 0 (conv: and every half-slab): all but the youngest two weight stages (+ acts)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    uint4 fwA[8], fwB[8], fxA[3], fxB[3];
    read_w(fwA, 0u, 0, 8);
    {
        const unsigned xa = xstage_of(0, 0, 0) + xoff_of(0);
        fxA[0] = *(const uint4*)(lds + xa);
        fxA[1] = *(const uint4*)(lds + xa + 1024);
    }
    // One K step: fwc / fxc hold this step's weight fragments and its first two activation fragments; the step reads its other six
    // activation fragments two row blocks ahead, requests one DMA piece per row block, passes the barrier behind block 5 and
    // reads the NEXT step's first fragments (fwn, fxn) under its last two blocks.
#define RG_STEP(fwc, fxc, fwn, fxn)                                                                                   \
    {                                                                                                                \
        const unsigned xa = xstage_of(q, cb, hf) + xoff_of(tap);                                                     \
        int nh = hf + 1, nt = tap + (nh >> 1), nc = cb;                                                              \
        nh &= 1;                                                                                                     \
        if (nt >= ntap) { nt = 0; ++nc; }                                                                            \
        const unsigned xn = xstage_of(q + 1, nc, nh) + xoff_of(nt);                                                  \
        const unsigned wnx = (unsigned)(((q + 1) & 3) * RG_WST);                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) {                                                           \
            if (mi + 2 < 8) fxc[(mi + 2) % 3] = *(const uint4*)(lds + xa + (mi + 2) * 1024);                         \
            if (mi == 6) read_w(fwn, wnx, 0, 4);                                                                     \
            if (mi == 7) {                                                                                           \
                read_w(fwn, wnx, 4, 8);                                                                              \
                fxn[0] = *(const uint4*)(lds + xn);                                                                  \
                fxn[1] = *(const uint4*)(lds + xn + 1024);                                                           \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            _Pragma("unroll") for (int ni = 0; ni < 8; ++ni) Mma16<T>::step(fwc[ni], fxc[mi % 3], acc[ni][mi]);      \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (RG_PROBE != 1 && mi < P) issue_piece(q, mi);                                                         \
            if (mi == 5) {                                                                                           \
                if (RG_PROBE == 0) RG_VMCNT(P + PB);                                                                 \
                asm volatile("" ::: "memory");                                                                       \
                __builtin_amdgcn_s_barrier();                                                                        \
                asm volatile("" ::: "memory");                                                                       \
            }                                                                                                        \
        }                                                                                                            \
        hf = nh; tap = nt; cb = nc;                                                                                  \
    }
#if RG_SINGLE
    for (int q = 0; q < nst; ++q) {
        RG_STEP(fwA, fxA, fwB, fxB)
#pragma unroll
        for (int i = 0; i < 8; ++i) fwA[i] = fwB[i];
        fxA[0] = fxB[0];
        fxA[1] = fxB[1];
    }
#else
    for (int q = 0; q < nst; q += 2) {  // (nst is even: two halves per 64-channel block)
        RG_STEP(fwA, fxA, fwB, fxB)
        {
            const int q1 = q + 1;
            const int q = q1;
            RG_STEP(fwB, fxB, fwA, fxA)
        }
    }
#endif
#undef RG_STEP
    RG_VMCNT(0);  // (the zero fills behind the last step)
    sbias[tid] = bias_v;  // (here, not in the prologue: hipcc would wait for the bias load with a vmcnt(0) - behind the prologue's requests)
    __syncthreads();

    // ---- epilogue: v = act(acc + bias), 8 consecutive channels per 16-byte store
    T* __restrict__ C = (T*)p.C + (size_t)ub * S * p.ldc;
    const bool fulln = n0 + RG_BN <= p.N;
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

bool gemm_ring_supported(const GemmArgs& a, int in_dtype, int out_dtype) {
    if (in_dtype != FS2_BF16 || out_dtype != FS2_BF16) return false;
    if (a.ln_g || a.dot_w || a.z_out || a.res || a.gate || a.zero_rows || a.C_lo || a.split || a.w_presplit || a.ksplit > 1 || a.drop_p > 0.f) return false;
    if (a.rs_stats || a.stats_out || a.epi_res || a.head_out || !a.C) return false;
    if (!(a.taps & 1) || a.taps > 31 || a.Cin % 64 || a.Cin < 128 || a.K != a.taps * a.Cin || a.N < 192 || a.N % 8) return false;
    const int S = a.taps == 1 ? a.M : a.S;
    if (S <= 0 || a.M % S) return false;
    if (a.ldx % 8 || a.ldc % 8) return false;
    if ((size_t)S * a.ldx * 2 >= 0xFFFFF000ull || (size_t)S * a.ldc * 2 >= 0xFFFFF000ull || (size_t)a.N * a.K * 2 >= 0xFFFFF000ull) return false;
    return true;
}

int launch_gemm_ring(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.taps == 1) a.S = a.M;
    a.pad = (a.taps - 1) / 2;
    const int tiles = (a.M / a.S) * ((a.S + RG_BM - 1) / RG_BM) * ((a.N + RG_BN - 1) / RG_BN);
    const dim3 g(tiles), b(256);
    if (a.taps == 1) hipLaunchKernelGGL((gemm_ring_kernel<true, 4>), g, b, 0, stream, a);
    else if (a.taps == 3) hipLaunchKernelGGL((gemm_ring_kernel<false, 3>), g, b, 0, stream, a);
    else if (a.taps < 9) hipLaunchKernelGGL((gemm_ring_kernel<false, 2>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((gemm_ring_kernel<false, 1>), g, b, 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
