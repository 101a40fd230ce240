// MFMA GEMM / implicit-GEMM Conv1d for gfx950.
//
//   C[m, n] = epilogue( sum_{tap, c} X[m + tap - pad, c] * W[n, tap*Cin + c] + bias[n] )
//
// replaces, on the FastSpeech2 forward path, every nn.Linear / pointwise Conv1d and the dense
// "same"-padded Conv1d of the reference (model.py:94-106 conv FFN, model.py:528-536 predictor
// conv, nn.MultiheadAttention in/out projections, fastspeech2.py:385-388 mel linear).
// Activations are (B*S, C) row-major; the zero "same" padding is applied per utterance (rows of
// different utterances are adjacent, the halo must not cross them) and is NOT masked by the
// padding mask, exactly as the reference's unmasked convs behave.
//
// Tiling: 128 (x rows) x 128 (W rows) per 256-thread workgroup, 4 waves as 2x2, each wave a
// 64x64 patch = 4x4 MFMA 16x16 fragments.  K advances in 128-byte chunks per row (64 bf16 /
// 32 fp32), double-buffered in LDS with an XOR-16B swizzle so every ds_read_b128 lane group is
// conflict free; the next chunk's global loads are issued before the current chunk's MFMAs and
// written to the other LDS buffer afterwards (one barrier per chunk).  MFMA operands are swapped
// (W is the row operand) so each lane ends up with 4 consecutive n of one output row: the
// epilogue is one 8/16-byte store per fragment.
#include <string.h>

#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

static constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB: bytes of K per LDS row
static constexpr int TILE_BYTES = BM * ROWB;           // 16 KiB per operand per buffer

__device__ inline int swz(int row, int slot) { return row * ROWB + ((slot ^ (row & 7)) << 4); }

template <typename T, typename OutT, bool ZR = false>
__global__ __launch_bounds__(256) void gemm_conv_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                   // [2][TILE_BYTES]
    unsigned char* sW = smem + 2 * TILE_BYTES;  // [2][TILE_BYTES]
    constexpr int E16 = Num<T>::kPer16B;        // elements per 16 bytes
    constexpr int KE = ROWB / (int)sizeof(T);   // k elements per chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;

    const T* __restrict__ X = (const T*)p.X;
    const T* __restrict__ W = (const T*)p.W;

    // staging assignment: 4 (row, slot) pairs per thread for each operand
    const int srow = tid >> 3, sslot = tid & 7;
    int xt[4];       // position of the row inside its utterance (conv zero padding)
    bool xin[4];     // row < M
    size_t xoff[4];  // element offset of (row, slot) at tap shift 0, channel 0
    size_t woff[4];
    bool win[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        xin[i] = m < p.M;
        xt[i] = (p.taps > 1) ? (m % p.S) : 0;
        xoff[i] = (size_t)m * p.ldx + sslot * E16;
        const int n = n0 + srow + 32 * i;
        win[i] = n < p.N;
        woff[i] = (size_t)n * p.K + sslot * E16;
    }

    const int nk = p.K / KE;
    uint4 rx[4], rw[4];

    auto load_chunk = [&](int kc) {
        const int k0 = kc * KE;
        int tap = 0, c0 = k0;
        if (p.taps > 1) { tap = k0 / p.Cin; c0 = k0 - tap * p.Cin; }
        const int shift = tap - p.pad;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool ok = xin[i];
            if (p.taps > 1) { const int t = xt[i] + shift; ok = ok && t >= 0 && t < p.S; }
            rx[i] = ok ? *(const uint4*)(X + xoff[i] + (ptrdiff_t)shift * p.ldx + c0) : make_uint4(0, 0, 0, 0);
            rw[i] = win[i] ? *(const uint4*)(W + woff[i] + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            *(uint4*)(sX + buf * TILE_BYTES + swz(row, sslot)) = rx[i];
            *(uint4*)(sW + buf * TILE_BYTES + swz(row, sslot)) = rw[i];
        }
    };

    f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_chunk(kc + 1);
        const unsigned char* bx = sX + buf * TILE_BYTES;
        const unsigned char* bw = sW + buf * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[4], fx[4];
            const int slot = ks * 4 + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fw[i] = *(const uint4*)(bw + swz(wn * 64 + i * 16 + fr, slot));
                fx[i] = *(const uint4*)(bx + swz(wm * 64 + i * 16 + fr, slot));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mma16<T>::step(fw[ni], fx[mi], acc[ni][mi]);
        }
        if (kc + 1 < nk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds, per fragment, out[m][n .. n+3]
    OutT* __restrict__ C = (OutT*)p.C;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + fg * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f}, wgv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) bv[r] = p.bias[n + r];
        }
        if (p.rs_stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) wgv[r] = p.rs_wg[n + r];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + fr;
            if (m >= p.M) continue;
            const bool zr = ZR && p.zero_rows[m];
            float2 rq = make_float2(1.f, 0.f);
            if (p.rs_stats) rq = ((const float2*)p.rs_stats)[m];  // the row-scaled product (GemmArgs::rs_stats): narrow heads (N < 192) take it here
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = p.rs_stats ? __builtin_fmaf(acc[ni][mi][r], rq.x, __builtin_fmaf(-rq.y, wgv[r], bv[r])) : acc[ni][mi][r] + bv[r];
                if (p.relu) v[r] = fmaxf(v[r], 0.f);
                if (ZR) v[r] = zr ? 0.f : v[r];
            }
            OutT* dst = C + (size_t)m * p.ldc + n;
            if (n + 3 < p.N) {
                if constexpr (sizeof(OutT) == 4) {
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *(uint2*)dst = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) dst[r] = Num<OutT>::from_f32(v[r]);
            }
        }
    }
}

// =================================================================================================
// Large-problem variant: 128 (x rows) x 256 (W rows) per 512-thread workgroup (8 waves as 2 x 4,
// each again a 64x64 patch), operands streamed global -> LDS by DMA (global_load_lds, 16 B per lane,
// no staging registers, XOR swizzle applied on the source address) through a 3-stage ring:
// chunk kc+2 is in flight while chunk kc is multiplied, with a counted vmcnt (never 0 in the steady
// state) and one raw s_barrier per chunk (cdna_hip_programming.md §5 "Pipelining across barriers").
// Rows that must read as zero (conv halo outside the utterance, M/N tails) are DMA'd from a zero
// page.  All LDS lives in ONE extern array (a second __shared__ object would make hipcc drain
// vmcnt before every ds_read).
// =================================================================================================
static constexpr int G2_BM = 128, G2_BN = 256;  // 3 LDS stages
static constexpr int G2_XB = G2_BM * ROWB;        // 16 KiB
static constexpr int G2_WB = G2_BN * ROWB;        // 32 KiB
static constexpr int G2_STAGE = G2_XB + G2_WB;    // 48 KiB
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

__device__ inline void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, typename OutT>
__global__ __launch_bounds__(512) void gemm_conv_glds_kernel(GemmArgs p) {
    // one array PER STAGE: hipcc tracks pending LDS-DMA writes per LDS object, so ds_reads of the
    // stage being multiplied do not wait for the DMAs still filling the other two
    __shared__ __attribute__((aligned(16))) unsigned char st0[G2_STAGE];  // [X 16K | W 32K]
    __shared__ __attribute__((aligned(16))) unsigned char st1[G2_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char st2[G2_STAGE];
    constexpr int E16 = Num<T>::kPer16B;
    constexpr int KE = ROWB / (int)sizeof(T);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + G2_BN - 1) / G2_BN;
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
    const int m0 = bm * G2_BM, n0 = bn * G2_BN;
    const T* __restrict__ X = (const T*)p.X;
    const T* __restrict__ W = (const T*)p.W;

    // DMA assignment: a wave instruction moves 64 x 16 B = 8 LDS rows.  X: 2 per wave, W: 4.
    // Exactly 6 DMA instructions per chunk per wave, unconditionally (the counted vmcnt below
    // relies on it): lanes whose row must read as zero point at the zero page instead of branching.
    int xt[2];            // row position inside its utterance (0 for plain GEMMs: always in range)
    uintptr_t xmask[2];   // all ones if row < M
    uintptr_t xaddr[2];   // byte address of (row, logical slot) at tap shift 0, channel 0
    uintptr_t waddr[4];   // byte address of (n, logical slot) at k = 0, or the zero page
    uintptr_t wstep[4];   // K advance in bytes (0 for zero-page lanes)
    const uintptr_t zaddr = (uintptr_t)g_zero_page;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int P = (i * 8 + wave) * 64 + lane, row = P >> 3, ps = P & 7;
        const int m = m0 + row;
        xmask[i] = (uintptr_t)0 - (uintptr_t)(m < p.M);
        xt[i] = (p.taps > 1) ? (m % p.S) : 0;
        xaddr[i] = (uintptr_t)(X + (size_t)m * p.ldx + (ps ^ (row & 7)) * E16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = (i * 8 + wave) * 64 + lane, row = P >> 3, ps = P & 7;
        const int n = n0 + row;
        const uintptr_t in = (uintptr_t)0 - (uintptr_t)(n < p.N);
        waddr[i] = ((uintptr_t)(W + (size_t)n * p.K + (ps ^ (row & 7)) * E16) & in) | (zaddr & ~in);
        wstep[i] = (uintptr_t)(KE * sizeof(T)) & in;
    }
    const int nk = p.K / KE;
    const int Seff = p.taps > 1 ? p.S : 1;
    int is_tap = 0, is_c0 = 0;  // (tap, channel offset) of the next chunk to be issued, in issue order
    const ptrdiff_t row_bytes = (ptrdiff_t)p.ldx * (ptrdiff_t)sizeof(T);

    auto issue = [&](unsigned char* sx) {  // chunks are issued strictly in order 0, 1, 2, ...
        const int shift = is_tap - p.pad;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = xt[i] + shift;
            const uintptr_t ok = xmask[i] & ((uintptr_t)0 - (uintptr_t)((t >= 0) & (t < Seff)));
            const uintptr_t a = xaddr[i] + (uintptr_t)(shift * row_bytes + is_c0 * (ptrdiff_t)sizeof(T));
            glds16((const void*)((a & ok) | (zaddr & ~ok)), sx + (i * 8 + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16((const void*)waddr[i], sx + G2_XB + (i * 8 + wave) * 1024);
            waddr[i] += wstep[i];
        }
        is_c0 += KE;
        if (is_c0 == p.Cin) { is_c0 = 0; ++is_tap; }
    };

    f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    auto compute = [&](const unsigned char* bx) {
        const unsigned char* bw = bx + G2_XB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[4], fx[4];
            const int slot = ks * 4 + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fw[i] = *(const uint4*)(bw + swz(wn * 64 + i * 16 + fr, slot));
                fx[i] = *(const uint4*)(bx + swz(wm * 64 + i * 16 + fr, slot));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mma16<T>::step(fw[ni], fx[mi], acc[ni][mi]);
        }
    };
    // one pipeline step: chunk kc (in `cur`) has landed once at most the 6 DMAs of chunk kc+1 are
    // still outstanding; the barrier also frees the stage chunk kc-1 was multiplied from, which is
    // where chunk kc+2 goes
#define FS2_G2_STEP(cur, nxt2, kcv)                                                   \
    {                                                                                 \
        if ((kcv) + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_s_barrier();                                                 \
        if ((kcv) + 2 < nk) issue(nxt2);                                               \
        compute(cur);                                                                 \
    }
    issue(st0);
    if (nk > 1) issue(st1);
    for (int kc = 0; kc < nk; kc += 3) {
        FS2_G2_STEP(st0, st2, kc)
        if (kc + 1 < nk) FS2_G2_STEP(st1, st0, kc + 1)
        if (kc + 2 < nk) FS2_G2_STEP(st2, st1, kc + 2)
    }
#undef FS2_G2_STEP

    OutT* __restrict__ C = (OutT*)p.C;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + fg * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) bv[r] = p.bias[n + r];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + fr;
            if (m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[ni][mi][r] + bv[r];
                if (p.relu) v[r] = fmaxf(v[r], 0.f);
            }
            OutT* dst = C + (size_t)m * p.ldc + n;
            if (n + 3 < p.N) {
                if constexpr (sizeof(OutT) == 4) {
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *(uint2*)dst = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) dst[r] = Num<OutT>::from_f32(v[r]);
            }
        }
    }
}

template <typename T, typename OutT>
static int launch_glds_t(const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + G2_BM - 1) / G2_BM) * ((a.N + G2_BN - 1) / G2_BN);
    hipLaunchKernelGGL((gemm_conv_glds_kernel<T, OutT>), dim3(tiles), dim3(512), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =================================================================================================
// Slab kernel: (MI*32) rows of ONE utterance x 256 output channels per 512-thread workgroup
// (8 waves as 2 x 4, wave patch (MI*16) x 64).  For a k-tap conv the x operand of all taps is the
// same (rows + k - 1) x 64-channel slab: it is DMA'd into LDS ONCE per channel block and the taps
// walk it with a row offset, so per output tile the activation traffic through L2 -> LDS drops
// k-fold; only the (256 x 64) weight tile changes every step.  Loop order: channel block (outer),
// tap (inner) - a pure re-association of the K sum.  Zero "same" padding = slab rows outside
// [0, S) of the utterance read the zero page.  Two slab buffers + two weight buffers, each its own
// LDS object; every step: explicit vmcnt(0) + __syncthreads, issue the
// next step's DMAs, multiply the current one.
// =================================================================================================
// Slab-kernel weight tile: MFMA row i (= lane group fg*4 + r in the accumulator) of fragment ni is
// mapped to output channel  (ni>>1)*32 + (i>>2)*8 + (ni&1)*4 + (i&3)  of the wave's 64, so that a lane
// ends up with 8 consecutive channels per fragment pair (wide, line-friendly stores).  wswz is the
// 16-byte XOR swizzle that keeps those fragment reads bank-conflict free.
__device__ inline int wcol(int ni, int fgq) { return (ni >> 1) * 32 + fgq * 8 + (ni & 1) * 4; }
__device__ inline int wswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

template <int MI> struct SlabCfg {
    static constexpr int BM = MI * 32;
    static constexpr int SLAB_GROUPS = (BM + 30 + 7) / 8;          // 8-row (1 KiB) DMA groups, k <= 31
    static constexpr int SI = (SLAB_GROUPS + 7) / 8;                // slab DMA instructions per wave
    static constexpr int SLAB_BYTES = SI * 8 * 1024;
};
static constexpr int S_BN = 256;
template <bool B> struct BoolC { static constexpr bool value = B; };

// (r02 built an in-place WIDE fused LayerNorm epilogue for N > 256 - one workgroup per row tile walks the column tiles, stores pre-norm
// rows, re-reads and normalises its own rows - which only tied two launches at M = 49152, K = 768 (118 vs 120 us) and lost elsewhere
// (K = 3072: 357 vs 294 us; M = 8192 / 12288: 1.5-2.2x slower); r04 an operand RING of 3 / 4 stages for pointwise launches on short
// tiles, measured neutral (these launches are bound by what a CU ingests per second, not by a step's round trip).  Both removed in r05;
// profiles/HISTORY.md §4 keeps the measurements.)
// SPLIT (T = float only): fp32 operands, bf16 x 3 arithmetic.  Each fp32 value is split in registers into a bf16 head
// and a bf16 tail (x = hi + lo up to 2^-17 |x|) and a product becomes three bf16 MFMAs, hi*hi + hi*lo + lo*hi, accumulated
// in fp32 (the dropped lo*lo term is 2^-16 of the product): ~1e-5 relative, ~400x closer to fp32 than bf16 storage, at
// 3 x 16-cycle MFMAs per 32 k-values instead of the 8 x 32-cycle fp32 MFMAs - the arithmetic of the "front" (encoder +
// variance adaptor) in the mixed precision mode, where a discrete decision hangs on every output.
// DEFER (LN = false only): the deferred-LayerNorm epilogue (residual preload + pre-norm store + row statistics); its own
// instantiation so that the plain GEMM launches do not carry its registers (the in-projection measured +9 % with both in one).
#ifdef FS2_SLAB_PROBE  // tools/probes/slab_phase_stamps.py: s_memtime of waves 0 and 7 of workgroups 0 and 128 at the phase boundaries
__device__ unsigned long long g_slab_stamps[4][8];
#define SLAB_STAMP(i)                                                                                                   \
    do {                                                                                                                \
        if ((blockIdx.x == 0 || blockIdx.x == 128) && (threadIdx.x == 0 || threadIdx.x == 448))                          \
            g_slab_stamps[(blockIdx.x ? 2 : 0) + (threadIdx.x ? 1 : 0)][i] = __builtin_amdgcn_s_memtime();               \
    } while (0)
#else
#define SLAB_STAMP(i) do { } while (0)
#endif

template <typename T, typename OutT, int MI, bool LN, bool SPLIT = false, bool DEFER = false, bool XPRE = false>
__global__ __launch_bounds__(512) void gemm_conv_slab_kernel(GemmArgs p) {
    static_assert(!XPRE || SPLIT, "XPRE: the split arithmetic's conv form (activation slab split in place when it lands)");
    static_assert(!SPLIT || sizeof(T) == 4, "the split arithmetic takes fp32 operands");
    static_assert(!(DEFER && LN), "deferred LayerNorm is what a launch WITHOUT the fused epilogue leaves behind");
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource builtins only exist in the device pass
    using Cfg = SlabCfg<MI>;
    constexpr int BMs = Cfg::BM;
    constexpr int SI = Cfg::SI;
    constexpr int SLAB_B = Cfg::SLAB_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char slab0[SLAB_B];
    __shared__ __attribute__((aligned(16))) unsigned char slab1[SLAB_B];
    __shared__ __attribute__((aligned(16))) unsigned char wt0[S_BN * ROWB];  // 32 KiB weight tile per stage
    __shared__ __attribute__((aligned(16))) unsigned char wt1[S_BN * ROWB];
    // The column tile's per-channel epilogue constants - bias, and for the fused LayerNorm epilogue gamma, beta, the predictor head's
    // weights - fetched at the TOP of the kernel (one value per thread, their round trip under the first operand DMAs') and parked
    // in LDS.  (Until r05 the epilogues fetched them behind the K loop: sixteen bias loads per lane, then gamma / beta - the
    // "barrier 3.1 k" and part of the "row statistics 4.5 k" of a 48 k-tick out-projection + LayerNorm launch,
    // tools/probes/slab_phase_stamps.py, with nothing to hide the two round trips under.)
    __shared__ __attribute__((aligned(16))) float sprm[(LN ? 4 : 1) * S_BN];
    constexpr int KE = ROWB / (int)sizeof(T);
    SLAB_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, nutt = p.M / S;
    const int tiles_n = (p.N + S_BN - 1) / S_BN;
    const int tiles_m = (S + BMs - 1) / BMs;
    int bid = blockIdx.x;
    if (p.xcd_remap == 2) {
        // Column PAIRS per XCD (the launcher checks the divisibilities): XCD x owns the column tiles {2 cg, 2 cg + 1}, cg = x % (tiles_n / 2),
        // and every (8 / (tiles_n / 2))-th row tile.  r05 counters on the decoder conv1 (4 column tiles, W = 4.7 MB, 4 MB of L2 per XCD): with
        // whole row tiles per XCD (mode 1) the slab is fetched once but every XCD streams the whole weight panel through its L2 once per
        // ROUND (25 + 113 MB fetched); with one column tile per XCD (what the plain order happens to give at 4 column tiles) the
        // weights stay (9 MB) and the slab is fetched four times (100 MB).  Pairs: 2.4 MB of weights per XCD stay resident, the slab
        // is fetched twice.
        const int ncg = tiles_n >> 1, xg = 8 / ncg, xcd = bid & 7, k = bid >> 3;
        bid = ((k >> 1) * xg + xcd / ncg) * tiles_n + (xcd % ncg) * 2 + (k & 1);
    } else if (p.xcd_remap) {
        // Workgroup i is dispatched to XCD i % 8.  Hand every XCD a CONTIGUOUS range of tiles, so that
        // the column tiles of one row tile (which read the same activation slab) run on the same XCD at
        // the same time and share it in that XCD's L2 instead of fetching it once per XCD.
        const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, xcd = bid & 7;
        bid = xcd * per + (xcd < rem ? xcd : rem) + (bid >> 3);
    }
    // Split-K launches (p.ksplit > 1; plain epilogue, OutT = float): split ks owns a contiguous range of the Cin / KE channel blocks (all
    // taps) and stores its partial sums into plane ks of C, (ksplit, M, ldc); split_k_reduce_kernel adds the planes up.  ks is the
    // SLOWEST tile index, so an XCD's contiguous tile range reads one or two weight slices, not the whole panel.
    int ks = 0;
    if constexpr (!LN && !DEFER) {
        if (p.ksplit > 1) {
            const int per_split = gridDim.x / p.ksplit;
            ks = bid / per_split;
            bid -= ks * per_split;
        }
    }
    const int bn = bid % tiles_n;
    bid /= tiles_n;
    const int tm = bid % tiles_m, ub = bid / tiles_m;
    if (ub >= nutt) return;
    const int t0 = tm * BMs;
    int n0 = bn * S_BN;
    const int ntap = p.taps, ncc = p.Cin / KE / (p.ksplit > 1 ? p.ksplit : 1), cc0 = ks * ncc;
    const T* __restrict__ Xu = (const T*)p.X + (size_t)ub * S * p.ldx + cc0 * KE;  // this utterance's rows (this split's channels)
    // Operands are fetched with buffer loads straight into LDS: a descriptor per operand in SGPRs,
    // one 32-bit byte offset per lane per DMA, the channel/tap advance in the scalar offset.  Lanes
    // whose row must read as zero (outside the utterance, M/N tails) carry an out-of-range offset:
    // the hardware bounds check returns zeros for them.
    constexpr unsigned OOB = 0xFFFFF000u;
    const unsigned xbytes = (unsigned)(((size_t)(S - 1) * p.ldx + p.Cin - cc0 * KE) * sizeof(T));
    const unsigned wbytes = (unsigned)((size_t)p.N * p.K * sizeof(T));
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)Xu, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, wbytes, 0x00020000);
    // DMA duty is NOT shared evenly: only waves 0 .. DW-1 issue the operand DMAs (2x the pieces each).  The issue
    // arbiter serves the oldest wave of a SIMD first, so waves 0-3 finish a step's MFMAs ~1000 cycles before
    // waves 4-7 and idle at the barrier, while an LDS-DMA instruction costs its wave 100-150 issue cycles next to
    // MFMAs (tools/probes/slab_stamps.py, step_shape_mfma.hip): the ~500 cycles of DMA issue per step belong on
    // the waves that have the slack, off the critical ones.
    constexpr int DW = 4, DSI = SI * 8 / DW, DWI = 32 / DW;
    const bool dma_wave = wave < DW;
    unsigned svoff[DSI];
#pragma unroll
    for (int i = 0; i < DSI; ++i) {
        const int P = (i * DW + (wave & (DW - 1))) * 64 + lane, row = P >> 3, ps = P & 7;
        const int t = t0 - p.pad + row;
        const bool ok = (t >= 0) & (t < S) & (row < BMs + ntap - 1);
        svoff[i] = ok ? (unsigned)t * (unsigned)(p.ldx * (int)sizeof(T)) + (unsigned)((ps ^ (row & 7)) << 4) : OOB;  // < 4 GiB (xbytes)
    }
    unsigned wvoff[DWI];
    auto set_wvoff = [&]() {
#pragma unroll
        for (int i = 0; i < DWI; ++i) {
            const int P = (i * DW + (wave & (DW - 1))) * 64 + lane, row = P >> 3, ps = P & 7;
            const int n = n0 + row;
            wvoff[i] = n < p.N ? (unsigned)n * (unsigned)(p.K * (int)sizeof(T)) + (unsigned)((ps ^ wswz(row)) << 4) : OOB;  // < 4 GiB (wbytes)
        }
    };
    set_wvoff();
    auto issue_slab = [&](unsigned char* dst, int cc) {
        if (!dma_wave) return;
#pragma unroll
        for (int i = 0; i < DSI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(dst + (i * DW + wave) * 1024),
                                                     16, svoff[i], cc * ROWB, 0, 0);
    };
    auto issue_w = [&](unsigned char* dst, int cc, int tap) {
        if (!dma_wave) return;
        const int koff = (tap * p.Cin + (cc0 + cc) * KE) * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < DWI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + (i * DW + wave) * 1024),
                                                     16, wvoff[i], koff, 0, 0);
    };

    f32x4_t acc[4][MI];  // [ni][mi]
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int xrow0 = wm * (MI * 16) + fr;  // slab row of fragment 0 at tap 0
    // Residual without an activation in front of it: start the accumulators AT the residual, fetched
    // with 16-byte loads that land underneath the first operand DMAs instead of sitting, exposed,
    // between the K loop and the row statistics.
    // (the deferred-LayerNorm epilogue does the same with ITS residual, normalised first if that is a pre-norm tensor)
    // (with the epilogue's dropout the residual must stay OUT of the accumulators: the mask applies to the product alone)
    const bool res_in_acc = !p.relu && !(LN && p.drop_p > 0.f) && (LN ? p.res != nullptr : (DEFER && p.epi_res != nullptr));
    int woff[4][2];                          // weight fragment byte offsets (tap independent)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int wrow = wn * 64 + wcol(i, fr >> 2) + (fr & 3);  // MFMA row fr of fragment i <-> channel
            woff[i][ks] = wrow * ROWB + (((ks * 4 + fg) ^ wswz(wrow)) << 4);
        }

    // SPLIT: the weights ARRIVE as heads + tails (slot fg of a row's 128-byte chunk = the bf16 heads of lane group fg's eight k-values,
    // slot 4 + fg their tails: presplit_pack_kernel, once per weight tensor); the activations are fp32 and are split either in
    // registers as a fragment is read (pointwise launches) or - XPRE, conv launches - in place when the slab lands (split_rows
    // below: once per element instead of once per column wave and tap).
    auto compute = [&](const unsigned char* sl, const unsigned char* wt, int tap) {
        if constexpr (SPLIT) {
            // both 16-byte chunks of the step = 8 k-values per lane = the k-group one bf16 16x16x32 MFMA takes from it
            uint4 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { wh[i] = *(const uint4*)(wt + woff[i][0]); wl[i] = *(const uint4*)(wt + woff[i][1]); }
            if constexpr (XPRE) {
                // no VALU between the fragment reads and the MFMAs: left alone hipcc hoists all 2 MI reads to the top (64 more live
                // registers next to 128 accumulators) - one row block ahead, pinned
                uint4 xh[2], xl[2];
                xh[0] = *(const uint4*)(sl + swz(xrow0 + tap, fg));
                xl[0] = *(const uint4*)(sl + swz(xrow0 + tap, 4 + fg));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (mi + 1 < MI) {
                        xh[(mi + 1) & 1] = *(const uint4*)(sl + swz(xrow0 + (mi + 1) * 16 + tap, fg));
                        xl[(mi + 1) & 1] = *(const uint4*)(sl + swz(xrow0 + (mi + 1) * 16 + tap, 4 + fg));
                    }
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        Mma16<bf16>::step(wl[ni], xh[mi & 1], acc[ni][mi]);  // small terms first
                        Mma16<bf16>::step(wh[ni], xl[mi & 1], acc[ni][mi]);
                        Mma16<bf16>::step(wh[ni], xh[mi & 1], acc[ni][mi]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    uint4 xh, xl;
                    split_bf16x3(*(const uint4*)(sl + swz(xrow0 + mi * 16 + tap, fg)), *(const uint4*)(sl + swz(xrow0 + mi * 16 + tap, 4 + fg)), xh, xl);
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        Mma16<bf16>::step(wl[ni], xh, acc[ni][mi]);  // small terms first
                        Mma16<bf16>::step(wh[ni], xl, acc[ni][mi]);
                        Mma16<bf16>::step(wh[ni], xh, acc[ni][mi]);
                    }
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[4], fx[MI];
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = *(const uint4*)(wt + woff[i][ks]);
#pragma unroll
            for (int i = 0; i < MI; ++i) fx[i] = *(const uint4*)(sl + swz(xrow0 + i * 16 + tap, ks * 4 + fg));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) Mma16<T>::step(fw[ni], fx[mi], acc[ni][mi]);
        }
        }
    };

    // SPLIT, conv launches: head / tail split of a landed fp32 activation slab, in place, ONCE per element: the pair of 16-byte slots
    // (s, 4 + s) of a row - the eight k-values lane group s feeds one bf16 MFMA - becomes (heads, tails).  Until r04 every wave split
    // every fragment it read: a weight value twice (2 row waves), an activation four times (4 column waves) and again at every conv
    // tap - (4 + MI) x ~24 VALU instructions per wave and step next to 12 MI MFMAs; the decoder conv1 ran 3.7x its bf16 launch.
    // Same arithmetic per element (split_bf16x3): bit-identical results.  (Splitting the weight tile in LDS as well - one more
    // barrier per step - measured slower on every pointwise launch: the weights are split once at load time instead.)
    auto split_rows = [&](unsigned char* buf, int nrows) {
        for (int i = tid; i < nrows * 4; i += 512) {
            const int row = i >> 2, sl4 = i & 3;
            const int x = row & 7;
            unsigned char* a0 = buf + row * ROWB + ((sl4 ^ x) << 4);
            unsigned char* a1 = buf + row * ROWB + (((4 + sl4) ^ x) << 4);
            uint4 h, l;
            split_bf16x3(*(const uint4*)a0, *(const uint4*)a1, h, l);
            *(uint4*)a0 = h;
            *(uint4*)a1 = l;
        }
    };
    // Step (cc, tap) uses slab[cc & 1] and wt[(cc*ntap + tap) & 1]; ntap is odd (the launcher checks),
    // so the weight parity is (cc + tap) & 1.  The steps are laid out as straight-line code with
    // STATIC buffer names (no pointer selects: hipcc then knows which LDS object each ds_read
    // touches and does not wait for the DMAs it has just issued into the other buffers).
#define FS2_SLAB_STEP(slab_cur, slab_nxt, w_cur, w_nxt, ccv, tapv)                          \
    {                                                                                       \
        dma_drain();     /* this wave's DMAs have landed ...                          */     \
        __syncthreads(); /* ... and so have everyone else's; the other buffers are free */  \
        int ntp = (tapv) + 1, ncb = (ccv);                                                  \
        if (ntp == ntap) { ntp = 0; ++ncb; }                                                \
        if (ncb < ncc) issue_w(w_nxt, ncb, ntp);                                            \
        if ((tapv) == 0 && (ccv) + 1 < ncc) issue_slab(slab_nxt, (ccv) + 1);                \
        if constexpr (XPRE) {                                                               \
            if ((tapv) == 0) {                                                              \
                split_rows(slab_cur, BMs + ntap - 1);                                       \
                __syncthreads();                                                            \
            }                                                                               \
        }                                                                                   \
        compute(slab_cur, w_cur, (tapv));                                                   \
    }
    // the first operand DMAs go out BEFORE the residual preload below: that preload waits for its own global loads (11 k of a
    // 35 k-tick out-projection + LayerNorm launch sat in front of the first DMA, tools/probes/slab_phase_stamps.py) and the two
    // round trips now overlap
    SLAB_STAMP(1);
    issue_slab(slab0, 0);
    issue_w(wt0, 0, 0);
    float prm[LN ? 4 : 1];
    if (tid < S_BN) {
        const int n = n0 + tid;
        const bool nv = n < p.N;
        prm[0] = (p.bias && nv) ? p.bias[n] : 0.f;
        if constexpr (LN) {  // (one column tile: n0 == 0)
            prm[1] = nv ? p.ln_g[n] : 0.f;
            prm[2] = nv ? p.ln_b[n] : 0.f;
            prm[3] = (nv && p.dot_w) ? p.dot_w[n] : 0.f;
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if ((LN || DEFER) && res_in_acc) {
        const T* R = (const T*)(LN ? p.res : p.epi_res) + (size_t)ub * S * p.ldc;
        const bool rnorm = DEFER && p.epi_res_stats != nullptr;
        const size_t rowbase0 = (size_t)ub * S;
        // Every load of the preload is issued before the first one is waited for.  (Until r04 each 16-byte load sat under its own
        // N-tail branch, and hipcc puts a vmcnt(0) behind a load it cannot move out of an exec-masked block: 2 * MI serial round
        // trips - twelve at 192-row tiles, the "residual preload 11.2 k ticks" of tools/probes/slab_phase_stamps.py.)  The tile
        // that lies wholly inside N - the common case - takes a branch-free path by workgroup-uniform dispatch.
        auto preload = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            // (groups of GP row blocks, one after the other, measured r04: GP = 2 at 192-row tiles costs the C3 step 1 ms - three
            // dependent round trips per workgroup in front of its first MFMA - and saves no register: the pressure was elsewhere)
            constexpr int GP = MI;
#pragma unroll
            for (int g0 = 0; g0 < MI; g0 += GP) {
            float rmean[GP], rrstd[GP];
            int tr[GP];
#pragma unroll
            for (int gi = 0; gi < GP; ++gi) {
                const int t = t0 + wm * (MI * 16) + (g0 + gi) * 16 + fr;
                tr[gi] = t >= S ? S - 1 : t;  // rows past the utterance end are never stored
                rmean[gi] = 0.f;
                rrstd[gi] = 1.f;
            }
            if (rnorm) {
                float2 pq[GP][4];
#pragma unroll
                for (int gi = 0; gi < GP; ++gi) {
                    const float2* ps = (const float2*)p.epi_res_stats + (rowbase0 + tr[gi]) * p.epi_res_parts;
#pragma unroll
                    for (int q = 0; q < 4; ++q) pq[gi][q] = ps[q < p.epi_res_parts ? q : 0];  // unconditional: a clamped index, masked below
                }
                const float invn = 1.0f / (float)p.N;
#pragma unroll
                for (int gi = 0; gi < GP; ++gi) {
#pragma unroll
                    for (int q = 1; q < 4; ++q) if (q >= p.epi_res_parts) pq[gi][q] = make_float2(0.f, 0.f);
                    const float s1 = (pq[gi][0].x + pq[gi][1].x) + (pq[gi][2].x + pq[gi][3].x);
                    const float s2 = (pq[gi][0].y + pq[gi][1].y) + (pq[gi][2].y + pq[gi][3].y);
                    rmean[gi] = s1 * invn;
                    rrstd[gi] = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-rmean[gi], rmean[gi], s2 * invn), 0.f) + p.ln_eps);
                }
            }
            if constexpr (FULL) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int gi = 0; gi < GP; ++gi)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const T* src = R + (size_t)tr[gi] * p.ldc + n0 + wn * 64 + j * 32 + fg * 8;
                            const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
                            acc[2 * j][g0 + gi] = (f32x4_t){q0.x, q0.y, q0.z, q0.w};
                            acc[2 * j + 1][g0 + gi] = (f32x4_t){q1.x, q1.y, q1.z, q1.w};
                        }
                } else {
                    uint4 rq[GP][2];
#pragma unroll
                    for (int gi = 0; gi < GP; ++gi)
#pragma unroll
                        for (int j = 0; j < 2; ++j) rq[gi][j] = *(const uint4*)(R + (size_t)tr[gi] * p.ldc + n0 + wn * 64 + j * 32 + fg * 8);
#pragma unroll
                    for (int gi = 0; gi < GP; ++gi)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const unsigned w4[4] = {rq[gi][j].x, rq[gi][j].y, rq[gi][j].z, rq[gi][j].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[2 * j + (e >> 1)][g0 + gi][(2 * e) & 3] = __uint_as_float(w4[e] << 16);
                                acc[2 * j + (e >> 1)][g0 + gi][(2 * e + 1) & 3] = __uint_as_float(w4[e] & 0xffff0000u);
                            }
                        }
                }
                if (rnorm) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = n0 + wn * 64 + j * 32 + fg * 8;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float gvr = p.epi_res_g[n + r], bvr = p.epi_res_b[n + r];
#pragma unroll
                            for (int gi = 0; gi < GP; ++gi)
                                acc[2 * j + (r >> 2)][g0 + gi][r & 3] = __builtin_fmaf((acc[2 * j + (r >> 2)][g0 + gi][r & 3] - rmean[gi]) * rrstd[gi], gvr, bvr);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int gi = 0; gi < GP; ++gi) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = n0 + wn * 64 + j * 32 + fg * 8;
                        const T* src = R + (size_t)tr[gi] * p.ldc + n;
                        float rv[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) rv[r] = (n + r < p.N) ? Num<T>::to_f32(src[r]) : 0.f;
                        if (rnorm) {
#pragma unroll
                            for (int r = 0; r < 8; ++r)
                                rv[r] = n + r < p.N ? __builtin_fmaf((rv[r] - rmean[gi]) * rrstd[gi], p.epi_res_g[n + r], p.epi_res_b[n + r]) : 0.f;
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) acc[2 * j + (r >> 2)][g0 + gi][r & 3] = rv[r];
                    }
                }
            }
            if (MI > GP) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (n0 + S_BN <= p.N) preload(BoolC<true>{});
        else preload(BoolC<false>{});
    }
    if (tid < S_BN) {  // (published by the first step's barrier; the epilogues read them after many more)
        sprm[tid] = prm[0];
        if constexpr (LN) {
            sprm[S_BN + tid] = prm[1];
            sprm[2 * S_BN + tid] = prm[2];
            sprm[3 * S_BN + tid] = prm[3];
        }
    }
    SLAB_STAMP(6);
    for (int cc = 0; cc < ncc; cc += 2) {
        for (int tap = 0; tap < ntap; tap += 2) {
            FS2_SLAB_STEP(slab0, slab1, wt0, wt1, cc, tap)
            if (tap + 1 < ntap) FS2_SLAB_STEP(slab0, slab1, wt1, wt0, cc, tap + 1)
        }
        if (cc + 1 < ncc) {
            for (int tap = 0; tap < ntap; tap += 2) {
                FS2_SLAB_STEP(slab1, slab0, wt1, wt0, cc + 1, tap)
                if (tap + 1 < ntap) FS2_SLAB_STEP(slab1, slab0, wt0, wt1, cc + 1, tap + 1)
            }
        }
    }
    SLAB_STAMP(2);
#undef FS2_SLAB_STEP

    if constexpr (LN) {
        // ---- fused row epilogue: the workgroup owns whole rows (tiles_n == 1) ----
        // LayerNorm(act(acc + bias) [+ res]) with two-pass statistics: lane partials -> lane-group
        // shuffles -> one LDS exchange between the four column waves; optional predictor head.
        const size_t rowbase = (size_t)ub * S;
        // On a SIMD the epilogue's VALU instructions and the MFMA passes of the other resident wave mostly ADD UP
        // (profiles/HISTORY.md §4, §7; for the K = 256 launches this epilogue is as many issue cycles as the K loop), so the
        // per-element selects that only matter for N < 256, for ReLU or for the predictor head are compiled out
        // of the common case by workgroup-uniform dispatch - same arithmetic, same order, same results.
        const bool full = p.N == S_BN;
        {   // ReLU as max(v, lo) with lo = 0 / -inf: one instruction either way, no per-element select
            const float lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = wn * 64 + wcol(ni, fg);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bvv = sprm[n + r];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[ni][mi][r] = fmaxf(acc[ni][mi][r] + bvv, lo);
                }
            }
            if (p.drop_p > 0.f) {
                // nn.Dropout between the sub-layer and its residual add (model.py:117-121: x = norm(x + dropout(sublayer(x)))): the
                // stand-alone kernel's mask (dropout_bits over the element index row * N + col of the (M, N) product), applied to the
                // fp32 value before the residual - the backward regenerates it from the same (seed, key)
                const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
                const float dsc = 1.f / (1.f - p.drop_p);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int t = t0 + wm * (MI * 16) + mi * 16 + fr;
                    const uint64_t e0 = (uint64_t)(rowbase + (t < S ? t : S - 1)) * (uint64_t)p.N;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const int n = wn * 64 + wcol(ni, fg);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[ni][mi][r] = dropout_bits(p.drop_seed, p.drop_key, e0 + (uint64_t)(n + r)) >= thr ? acc[ni][mi][r] * dsc : 0.f;
                    }
                }
            }
            if (!full) {  // columns past N hold act(0 + 0): make them exact zeros for the row sums
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = wn * 64 + wcol(ni, fg);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r >= p.N) {
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) acc[ni][mi][r] = 0.f;
                        }
                }
            }
        }
        if (p.res && !res_in_acc) {
            const T* R = (const T*)p.res;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int t = t0 + wm * (MI * 16) + mi * 16 + fr;
                if (t >= S) t = S - 1;  // rows past the utterance end are never stored
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = wn * 64 + wcol(ni, fg);
                    if (n + 3 < p.N) {
                        float rv[4];
                        if constexpr (sizeof(T) == 4) {
                            const float4 q = *(const float4*)(R + (rowbase + t) * p.ldc + n);
                            rv[0] = q.x; rv[1] = q.y; rv[2] = q.z; rv[3] = q.w;
                        } else {
                            const uint2 q = *(const uint2*)(R + (rowbase + t) * p.ldc + n);
                            rv[0] = __uint_as_float(q.x << 16); rv[1] = __uint_as_float(q.x & 0xffff0000u);
                            rv[2] = __uint_as_float(q.y << 16); rv[3] = __uint_as_float(q.y & 0xffff0000u);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[ni][mi][r] += rv[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) acc[ni][mi][r] += Num<T>::to_f32(R[(rowbase + t) * p.ldc + n + r]);
                    }
                }
            }
        }
        if (p.z_out) {  // training tape: the pre-norm rows, in the output dtype, 8 consecutive channels per store
            OutT* __restrict__ Zn = (OutT*)p.z_out + rowbase * p.ldc;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int t = t0 + wm * (MI * 16) + mi * 16 + fr;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = wn * 64 + j * 32 + fg * 8;
                    if (t < S && n < p.N) {
                        OutT* dst = Zn + (size_t)t * p.ldc + n;
                        const f32x4_t a0 = acc[2 * j][mi], a1 = acc[2 * j + 1][mi];
                        if (n + 7 < p.N) {
                            if constexpr (sizeof(OutT) == 4) {
                                *(float4*)dst = make_float4(a0[0], a0[1], a0[2], a0[3]);
                                *(float4*)(dst + 4) = make_float4(a1[0], a1[1], a1[2], a1[3]);
                            } else {
                                *(uint4*)dst = make_uint4(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]),
                                                          pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3]));
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 8; ++r)
                                if (n + r < p.N) dst[r] = Num<OutT>::from_f32(r < 4 ? a0[r] : a1[r - 4]);
                        }
                    }
                }
            }
        }
        __syncthreads();  // every wave is done with the operand buffers: reuse them for the exchange
        SLAB_STAMP(3);
        float* red = (float*)slab0;  // [4 column waves][BMs rows]
        const float* lnp = sprm + S_BN;  // [gamma 256 | beta 256 | head weight 256], parked at the top of the kernel
        const float invn = 1.0f / (float)p.N;
        float mean[MI], rstd[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float sm = 0.f;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) sm += (acc[ni][mi][0] + acc[ni][mi][1]) + (acc[ni][mi][2] + acc[ni][mi][3]);
            sm = group4_sum(sm);
            if (fg == 0) red[wn * BMs + wm * (MI * 16) + mi * 16 + fr] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * (MI * 16) + mi * 16 + fr;
            mean[mi] = ((red[row] + red[BMs + row]) + (red[2 * BMs + row] + red[3 * BMs + row])) * invn;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)  // acc <- acc - mean once: the variance and the normalisation both use it
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ni][mi][r] -= mean[mi];
        auto sq_dev = [&](auto full_c) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float q = 0.f;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = wn * 64 + wcol(ni, fg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = (decltype(full_c)::value || n + r < p.N) ? acc[ni][mi][r] : 0.f;
                        q = __builtin_fmaf(d, d, q);  // explicit: an SLP-packed mul + add would round differently per variant
                    }
                }
                q = group4_sum(q);
                if (fg == 0) red[wn * BMs + wm * (MI * 16) + mi * 16 + fr] = q;
            }
        };
        if (full) sq_dev(BoolC<true>{});
        else sq_dev(BoolC<false>{});
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * (MI * 16) + mi * 16 + fr;
            const float var = ((red[row] + red[BMs + row]) + (red[2 * BMs + row] + red[3 * BMs + row])) * invn;
            rstd[mi] = 1.0f / sqrtf(var + p.ln_eps);
        }
        OutT* __restrict__ Cn = p.C ? (OutT*)p.C + rowbase * p.ldc : nullptr;
        SLAB_STAMP(4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int t = t0 + wm * (MI * 16) + mi * 16 + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = wn * 64 + j * 32 + fg * 8;  // 8 consecutive channels of fragments 2j, 2j+1
                float y[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 g4 = *(const float4*)(lnp + n + 4 * h), b4 = *(const float4*)(lnp + S_BN + n + 4 * h);
                    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * h + r] = __builtin_fmaf(acc[2 * j + h][mi][r] * rstd[mi], gg[r], bb[r]);
                }
                if (Cn && t < S && n < p.N) {
                    OutT* dst = Cn + (size_t)t * p.ldc + n;
                    if (n + 7 < p.N) {
                        if constexpr (sizeof(OutT) == 4) {
                            *(float4*)dst = make_float4(y[0], y[1], y[2], y[3]);
                            *(float4*)(dst + 4) = make_float4(y[4], y[5], y[6], y[7]);
                        } else {
                            *(uint4*)dst = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]),
                                                      pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7]));
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) if (n + r < p.N) dst[r] = Num<OutT>::from_f32(y[r]);
                    }
                }
            }
        }
        SLAB_STAMP(5);
        if (p.dot_w) {  // predictor head (rare): the normalised values once more, times the head weights
            float dsum[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                dsum[mi] = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = wn * 64 + j * 32 + fg * 8;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 g4 = *(const float4*)(lnp + n + 4 * h), b4 = *(const float4*)(lnp + S_BN + n + 4 * h);
                        const float4 w4 = *(const float4*)(lnp + 2 * S_BN + n + 4 * h);
                        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
                        const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            dsum[mi] = __builtin_fmaf(__builtin_fmaf(acc[2 * j + h][mi][r] * rstd[mi], gg[r], bb[r]), ww[r], dsum[mi]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float d = dsum[mi];
                d = group4_sum(d);
                if (fg == 0) red[wn * BMs + wm * (MI * 16) + mi * 16 + fr] = d;
            }
            __syncthreads();
            if (wn == 0 && fg == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * (MI * 16) + mi * 16 + fr, t = t0 + row;
                    if (t < S) {
                        const float d = ((red[row] + red[BMs + row]) + (red[2 * BMs + row] + red[3 * BMs + row])) + p.dot_b;
                        p.pred[rowbase + t] = (p.mask && p.mask[rowbase + t]) ? 0.f : d;
                    }
                }
            }
        }
        return;
    }
    OutT* __restrict__ C = (OutT*)p.C + ((size_t)ks * p.M + (size_t)ub * S) * p.ldc;
    // lane (fr, fg) holds, for output row t, the 8 consecutive channels n .. n+7 of each fragment
    // pair (2j, 2j+1): one 16-byte (bf16) / two 16-byte (fp32) stores, 64 contiguous bytes per row
    // across the four lane groups; the two halves (j = 0, 1) of a row's 128-byte line go out back to back
    float bv[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fg * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[j][r] = sprm[n - n0 + r];
    }
    // ReLU and the N-tail checks are compiled out of the common case by workgroup-uniform dispatch (see the
    // LayerNorm epilogue above): at K = 256 this store loop is as many issue cycles as the K loop
    // gate_c: out = gate[row][col] > 0 ? v : 0 with `gate` a tensor of C's layout and dtype - the ReLU backward of a data-gradient
    // product (training step: dh = (dc2 . W2) o [h > 0]) folded into the store instead of a 3-tensor elementwise pass
    auto store = [&](auto relu_c, auto full_c, auto gate_c, auto drop_c, auto rs_c) {
    // rs_c: the row-scaled product (GemmArgs::rs_stats): v = rstd * acc - (rstd * mean) * wg + bias'
    float rsr[decltype(rs_c)::value ? MI : 1], rsm[decltype(rs_c)::value ? MI : 1], wgv[decltype(rs_c)::value ? 2 : 1][8];
    if constexpr (decltype(rs_c)::value) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int t = t0 + wm * (MI * 16) + mi * 16 + fr;
            const float2 q = ((const float2*)p.rs_stats)[(size_t)ub * S + (t < S ? t : S - 1)];
            rsr[mi] = q.x;
            rsm[mi] = q.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fg * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) wgv[j][r] = n + r < p.N ? p.rs_wg[n + r] : 0.f;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int t = t0 + wm * (MI * 16) + mi * 16 + fr;
        if (t >= S) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fg * 8;
            if (!decltype(full_c)::value && n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if constexpr (decltype(rs_c)::value) v[r] = __builtin_fmaf(acc[2 * j + (r >> 2)][mi][r & 3], rsr[mi], __builtin_fmaf(-rsm[mi], wgv[j][r], bv[j][r]));
                else v[r] = acc[2 * j + (r >> 2)][mi][r & 3] + bv[j][r];
                if constexpr (decltype(relu_c)::value) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (decltype(drop_c)::value) {
                // nn.Dropout behind the activation (the FFN's hidden tensor in training mode, model.py:94-106): fs2_op_dropout's mask over the
                // (M, N) product, in the store instead of a read-modify-write pass over the (M, filter) tensor
                const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
                const float dsc = 1.f / (1.f - p.drop_p);
                const uint64_t e0 = ((uint64_t)ub * S + (uint64_t)t) * (uint64_t)p.N + (uint64_t)n;
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = dropout_bits(p.drop_seed, p.drop_key, e0 + r) >= thr ? v[r] * dsc : 0.f;
            }
            if constexpr (decltype(gate_c)::value) {
                const OutT* gt = (const OutT*)((const char*)p.gate + ((size_t)ub * S * p.ldc + (size_t)t * p.ldc + n) * sizeof(OutT));
                if (decltype(full_c)::value || n + 7 < p.N) {
                    float gv[8];
                    if constexpr (sizeof(OutT) == 4) {
                        const float4 q0 = *(const float4*)gt, q1 = *(const float4*)(gt + 4);
                        gv[0] = q0.x; gv[1] = q0.y; gv[2] = q0.z; gv[3] = q0.w; gv[4] = q1.x; gv[5] = q1.y; gv[6] = q1.z; gv[7] = q1.w;
                    } else {
                        const uint4 q = *(const uint4*)gt;
                        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gv[2 * e] = __uint_as_float(w4[e] << 16);
                            gv[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = gv[r] > 0.f ? v[r] * p.gate_scale : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = (n + r < p.N && Num<OutT>::to_f32(gt[r]) > 0.f) ? v[r] * p.gate_scale : 0.f;
                }
            }
            if constexpr (sizeof(OutT) == 4) {
                if (p.C_lo) {  // the fp32 result as two bf16 tensors, head + tail (the split-arithmetic attention's operands): same bytes
                    const size_t off = ((size_t)ub * S + t) * p.ldc + n;
                    uint4 hq, lq;
                    split_bf16x3(make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])),
                                 make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])), hq, lq);
                    bf16* dh = (bf16*)p.C + off;
                    bf16* dl = (bf16*)p.C_lo + off;
                    if (decltype(full_c)::value || n + 7 < p.N) {
                        *(uint4*)dh = hq;
                        *(uint4*)dl = lq;
                    } else {
                        const unsigned hw[4] = {hq.x, hq.y, hq.z, hq.w}, lw[4] = {lq.x, lq.y, lq.z, lq.w};
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            if (n + r < p.N) {
                                ((unsigned short*)dh)[r] = (unsigned short)(hw[r >> 1] >> (16 * (r & 1)));
                                ((unsigned short*)dl)[r] = (unsigned short)(lw[r >> 1] >> (16 * (r & 1)));
                            }
                    }
                    continue;
                }
            }
            OutT* dst = (OutT*)((char*)C + (unsigned)(t * p.ldc + n) * (unsigned)sizeof(OutT));  // one utterance < 4 GiB
            if (decltype(full_c)::value || n + 7 < p.N) {
                if constexpr (sizeof(OutT) == 4) {
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    *(uint4*)dst = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                              pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (n + r < p.N) dst[r] = Num<OutT>::from_f32(v[r]);
            }
        }
    }
    };
    if constexpr (DEFER) {
        // Deferred-LayerNorm epilogue (rows wider than one tile): v = act(acc + bias) + res, stored as it is, plus this
        // column tile's sum(v), sum(v^2) per row -> stats_out[row][column tile]; the consumers (depth-wise conv,
        // normalise-only LayerNorm, the next epilogue's residual) finish mean / rstd from the tiles' parts.
        // The residual may itself be such a pre-norm tensor: then it is normalised on load from ITS parts.
        const size_t rowbase = (size_t)ub * S;
        const float lo = p.relu ? 0.f : -__builtin_inff();
        float a1[MI], a2[MI];
        auto body = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int t = t0 + wm * (MI * 16) + mi * 16 + fr;
            a1[mi] = a2[mi] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + fg * 8;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(acc[2 * j + (r >> 2)][mi][r & 3] + bv[j][r], lo);
                // (a residual reaches this epilogue inside the accumulators only: res_in_acc; one behind a ReLU the launcher refuses -
                //  the in-epilogue path cost 35 registers at 192-row tiles and was dead code in every launch of the forward)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (!FULL && n + r >= p.N) v[r] = 0.f;
                    a1[mi] += v[r];
                    a2[mi] = __builtin_fmaf(v[r], v[r], a2[mi]);
                }
                if (t < S && (FULL || n < p.N)) {
                    OutT* dst = (OutT*)((char*)C + (unsigned)(t * p.ldc + n) * (unsigned)sizeof(OutT));
                    if (FULL || n + 7 < p.N) {
                        if constexpr (sizeof(OutT) == 4) {
                            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                            *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        } else {
                            *(uint4*)dst = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                      pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) if (n + r < p.N) dst[r] = Num<OutT>::from_f32(v[r]);
                    }
                }
            }
        }
        };
        if (n0 + S_BN <= p.N) body(BoolC<true>{});
        else body(BoolC<false>{});
        if (p.stats_out) {  // one (sum, sum of squares) per row per column TILE: the four column waves meet in LDS
            __syncthreads();  // every wave is done with the operand buffers
            float2* red = (float2*)slab0;  // [4 column waves][BMs rows]
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const float s1 = group4_sum(a1[mi]), s2 = group4_sum(a2[mi]);
                if (fg == 0) red[wn * BMs + wm * (MI * 16) + mi * 16 + fr] = make_float2(s1, s2);
            }
            __syncthreads();
            for (int row = tid; row < BMs; row += 512) {
                const int t = t0 + row;
                if (t < S) {
                    const float2 q0 = red[row], q1 = red[BMs + row], q2 = red[2 * BMs + row], q3 = red[3 * BMs + row];
                    ((float2*)p.stats_out)[(rowbase + t) * (size_t)tiles_n + bn] =
                        make_float2((q0.x + q1.x) + (q2.x + q3.x), (q0.y + q1.y) + (q2.y + q3.y));
                }
            }
        }
        return;
    }
    const bool fulln = n0 + S_BN <= p.N;  // this column tile lies wholly inside N
    if (p.rs_stats) {  // (the launcher admits the row-scaled product without ReLU / gate / dropout / head + tail store, bf16 only)
        if constexpr (sizeof(T) == 2 && sizeof(OutT) == 2 && !SPLIT) {
            if (fulln) store(BoolC<false>{}, BoolC<true>{}, BoolC<false>{}, BoolC<false>{}, BoolC<true>{});
            else store(BoolC<false>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{}, BoolC<true>{});
        }
    } else if (p.gate) {  // (the launcher admits a gate without ReLU only)
        if (fulln) store(BoolC<false>{}, BoolC<true>{}, BoolC<true>{}, BoolC<false>{}, BoolC<false>{});
        else store(BoolC<false>{}, BoolC<false>{}, BoolC<true>{}, BoolC<false>{}, BoolC<false>{});
    } else if (p.drop_p > 0.f) {  // (the launcher admits the store's dropout with ReLU, bf16 / fp32 in = out, no gate)
        if (fulln) store(BoolC<true>{}, BoolC<true>{}, BoolC<false>{}, BoolC<true>{}, BoolC<false>{});
        else store(BoolC<true>{}, BoolC<false>{}, BoolC<false>{}, BoolC<true>{}, BoolC<false>{});
    } else if (fulln) {
        if (p.relu) store(BoolC<true>{}, BoolC<true>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{});
        else store(BoolC<false>{}, BoolC<true>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{});
    } else {
        if (p.relu) store(BoolC<true>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{});
        else store(BoolC<false>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{}, BoolC<false>{});
    }
#else
    (void)p;
#endif
}

template <typename T, typename OutT, int MI, bool LN, bool SPLIT = false, bool DEFER = false>
static int launch_slab_t(const GemmArgs& a0, hipStream_t stream) {
    GemmArgs a = a0;
    a.xcd_remap = tuning_of(a0.tune).slab_xcd_remap;
    const int BMs = SlabCfg<MI>::BM;
    if (a.xcd_remap == 2) {  // column pairs per XCD: 4 / 8 / 16 column tiles, a weight panel that does not fit an XCD's L2, whole rounds of row tiles
        const int tn = (a.N + S_BN - 1) / S_BN, rows = (a.M / a.S) * ((a.S + BMs - 1) / BMs);
        const bool ok = (tn == 4 || tn == 8 || tn == 16) && a.ksplit <= 1 && rows % (16 / tn) == 0 &&
                        (size_t)a.N * a.K * sizeof(T) > (size_t)3 << 20;
        if (!ok) a.xcd_remap = 1;
    }
    if constexpr (SPLIT) {  // conv launches of the split arithmetic: the slab is split in place when it lands
        if (a.taps > 1) {
            const int tiles = (a.M / a.S) * ((a.S + BMs - 1) / BMs) * ((a.N + S_BN - 1) / S_BN) * (a.ksplit > 1 ? a.ksplit : 1);
            hipLaunchKernelGGL((gemm_conv_slab_kernel<T, OutT, MI, LN, SPLIT, DEFER, true>), dim3(tiles), dim3(512), 0, stream, a);
            return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
        }
    }
    if (a.taps == 1) a.S = a.M;  // plain GEMM: one "utterance" of M rows
    const int tiles = (a.M / a.S) * ((a.S + BMs - 1) / BMs) * ((a.N + S_BN - 1) / S_BN) * (a.ksplit > 1 ? a.ksplit : 1);
    hipLaunchKernelGGL((gemm_conv_slab_kernel<T, OutT, MI, LN, SPLIT, DEFER>), dim3(tiles), dim3(512), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// Split launches whose caller holds plain fp32 weights (the operator-level entry points with Tuning::split_f32, knob 501: tests
// only - an engine packs its weights once at fs2_finalize, GemmArgs::w_presplit): packed on the fly into a block that lives for
// this launch, allocated and freed in STREAM ORDER (no process-wide buffer: until r05 one static grow-only block was shared by
// every caller and stream).  Not under stream capture.
static int presplit_on_the_fly(GemmArgs& a, hipStream_t stream, void** tmp) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return FS2_ERR_STATE;
    const size_t bytes = (size_t)a.N * a.K * 4;
    if (hipMallocAsync(tmp, bytes, stream) != hipSuccess) return FS2_ERR_NOMEM;
    const int r = launch_presplit_pack((const float*)a.W, *tmp, (size_t)a.N * a.K, stream);
    a.W = *tmp;
    a.w_presplit = 1;
    return r;
}

template <int MI>
static int launch_slab_do(const GemmArgs& a, int in_dtype, int out_dtype, bool sp, hipStream_t stream) {
    if (a.ln_g) {  // fused LayerNorm epilogue: whole rows per workgroup (N <= 256), tile heights up to 192
        if constexpr (MI <= 6) {
            if (a.N > S_BN) return FS2_ERR_SHAPE;
            if (in_dtype == FS2_F32 && out_dtype == FS2_F32)
                return sp ? launch_slab_t<float, float, MI, true, true>(a, stream) : launch_slab_t<float, float, MI, true>(a, stream);
            if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_slab_t<bf16, bf16, MI, true>(a, stream);
        }
        return FS2_ERR_SHAPE;
    }
    if (a.stats_out || a.epi_res) {  // deferred-LayerNorm epilogue (tile heights up to 192: the 256-row form spills)
        if (a.epi_res && a.relu) return FS2_ERR_SHAPE;  // the residual rides in the accumulators' initial value: no activation in between
        if constexpr (MI <= 6) {
            if (in_dtype == FS2_F32 && out_dtype == FS2_F32)
                return sp ? launch_slab_t<float, float, MI, false, true, true>(a, stream) : launch_slab_t<float, float, MI, false, false, true>(a, stream);
            if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_slab_t<bf16, bf16, MI, false, false, true>(a, stream);
        }
        return FS2_ERR_SHAPE;
    }
    if (in_dtype == FS2_F32 && out_dtype == FS2_F32)
        return sp ? launch_slab_t<float, float, MI, false, true>(a, stream) : launch_slab_t<float, float, MI, false>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_slab_t<bf16, bf16, MI, false>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_slab_t<bf16, float, MI, false>(a, stream);
    return FS2_ERR_SHAPE;
}

template <int MI>
static int launch_slab(const GemmArgs& a_in, int in_dtype, int out_dtype, hipStream_t stream) {
    GemmArgs a = a_in;
    const bool sp = a.split || tuning_of(a.tune).split_f32;
    void* tmp = nullptr;
    if (in_dtype == FS2_F32 && out_dtype == FS2_F32 && sp && !a.w_presplit) {
        const int r = presplit_on_the_fly(a, stream, &tmp);
        if (r != FS2_OK) { if (tmp) (void)hipFreeAsync(tmp, stream); return r; }
    }
    const int r = launch_slab_do<MI>(a, in_dtype, out_dtype, sp, stream);
    if (tmp) (void)hipFreeAsync(tmp, stream);  // behind the launch that reads it, in stream order
    return r;
}

template <typename T, typename OutT, bool ZR = false>
static int launch_t(const GemmArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    const size_t smem = 4 * TILE_BYTES;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_conv_kernel<T, OutT, ZR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return FS2_ERR_HIP;
        attr_set = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_conv_kernel<T, OutT, ZR>), dim3(tiles), dim3(256), smem, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

static int launch_gemm_plain(const GemmArgs& a, int in_dtype, int out_dtype, hipStream_t stream, bool* fused);

int launch_gemm(const GemmArgs& a, int in_dtype, int out_dtype, hipStream_t stream) {
    const Tuning& tn = tuning_of(a.tune);
    const int g_gemm_variant = tn.gemm_variant, g_gemm_wres = tn.gemm_wres;
    if (a.head_out) {  // head sums instead of the rows: the persistent kernel's deferred epilogue only (callers ask gemm_head_supported first)
        if (a.ln_g || !gemm_head_supported(a, in_dtype, out_dtype)) return FS2_ERR_SHAPE;
        return launch_gemm_persist(a, 6, stream);
    }
    if (!a.ln_g && a.drop_p > 0.f) {  // the plain store's dropout: slab kernel, behind a ReLU (the FFN's hidden tensor)
        if (!a.relu || a.gate || a.stats_out || a.epi_res || a.ksplit > 1 || a.zero_rows || g_gemm_variant != 0 || a.N < 192 || a.M % a.S ||
            !(a.taps & 1) || in_dtype != out_dtype)
            return FS2_ERR_SHAPE;
        return launch_gemm_plain(a, in_dtype, out_dtype, stream, nullptr);  // (FS2_ERR_SHAPE there if it would take a flat kernel)
    }
    if (!a.ln_g) {
        // K = 256 bf16: the column tile's weights live in registers, row tiles stream (gemm_wres.hip; bit-identical results)
        if (g_gemm_wres && g_gemm_variant == 0 && a.ksplit <= 1 && gemm_wres_supported(a, in_dtype, out_dtype, g_gemm_wres == 2)) return launch_gemm_wres(a, stream);
        return launch_gemm_plain(a, in_dtype, out_dtype, stream, nullptr);
    }
    // fused row epilogue requested: try the slab kernel (whole rows per workgroup), else GEMM -> ln_tmp
    // followed by the stand-alone LayerNorm kernel (same arithmetic, one more HBM round trip)
    bool fused = false;
    if (a.z_out && a.N > S_BN) return FS2_ERR_SHAPE;  // the pre-norm store exists in the one-column-tile epilogue only
    if (a.drop_p > 0.f && (a.N > S_BN || !a.z_out)) return FS2_ERR_SHAPE;  // the epilogue's dropout: one-column-tile fused form only (training tape)
    if (a.N <= S_BN && in_dtype == out_dtype) {
        const int r = launch_gemm_plain(a, in_dtype, out_dtype, stream, &fused);
        if (r != FS2_OK || fused) return r;
    }
    if (a.z_out) return FS2_ERR_SHAPE;  // not fusable for this shape: the caller takes its two-launch path (nothing was launched)
    if (!a.ln_tmp || in_dtype != out_dtype) return FS2_ERR_ARG;
    GemmArgs g = a;
    g.ln_g = nullptr;
    g.C = a.ln_tmp;
    const int r = launch_gemm_plain(g, in_dtype, out_dtype, stream, nullptr);
    if (r != FS2_OK) return r;
    LayerNormArgs l;
    l.x = a.ln_tmp; l.res = a.res; l.gamma = a.ln_g; l.beta = a.ln_b; l.y = a.C;
    l.dot_w = a.dot_w; l.dot_b = a.dot_b; l.mask = a.mask; l.pred = a.pred;
    l.M = a.M; l.H = a.N; l.eps = a.ln_eps;
    return launch_layernorm(l, out_dtype, stream);
}

// fused != nullptr: the caller wants the LN epilogue; only the slab kernel provides it.  If the slab
// kernel is not selected, nothing is launched and *fused stays false.
static int launch_gemm_plain(const GemmArgs& a, int in_dtype, int out_dtype, hipStream_t stream, bool* fused) {
    const Tuning& tn_ = tuning_of(a.tune);
    const int g_gemm_variant = tn_.gemm_variant;
    const bool g_split_f32 = tn_.split_f32 != 0;
    if (a.M <= 0 || a.N <= 0) return FS2_OK;
    const int ke = in_dtype == FS2_BF16 ? 64 : 32;
    const int e16 = in_dtype == FS2_BF16 ? 8 : 4;
    if (a.K % ke || a.Cin % ke || a.ldx % e16 || a.K != a.taps * a.Cin) return FS2_ERR_SHAPE;
    if (a.ldc % 4) return FS2_ERR_SHAPE;
    if (a.rs_stats && a.N < 192) {  // a narrow head behind a folded LayerNorm (the mel Linear): 128x128 kernel, bf16 operands
        if (fused || a.relu || a.gate || a.stats_out || a.epi_res || a.C_lo || a.ksplit > 1 || a.drop_p > 0.f || !a.rs_wg || a.taps != 1 || in_dtype != FS2_BF16)
            return FS2_ERR_SHAPE;
        if (out_dtype == FS2_F32) return a.zero_rows ? launch_t<bf16, float, true>(a, stream) : launch_t<bf16, float>(a, stream);
        if (out_dtype == FS2_BF16 && !a.zero_rows) return launch_t<bf16, bf16>(a, stream);
        return FS2_ERR_SHAPE;
    }
    if (a.zero_rows) {  // the mel head with zeroed pad rows: 128x128 kernel only
        if (fused) return FS2_ERR_SHAPE;
        if (in_dtype == FS2_F32 && out_dtype == FS2_F32) return launch_t<float, float, true>(a, stream);
        if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_t<bf16, float, true>(a, stream);
        return FS2_ERR_SHAPE;
    }
    if (a.rs_stats && (fused || a.relu || a.gate || a.stats_out || a.epi_res || a.zero_rows || a.C_lo || a.ksplit > 1 || a.drop_p > 0.f || g_gemm_variant == 1 ||
                       g_gemm_variant == 2 || !a.rs_wg || !(a.M % a.S == 0 && (a.taps & 1)) || a.N < 192 || in_dtype != FS2_BF16 ||
                       out_dtype != FS2_BF16))
        return FS2_ERR_SHAPE;  // the row-scaled product lives in the slab / persistent kernels' plain bf16 epilogue only
    if (a.gate && (fused || a.relu || a.stats_out || a.epi_res || g_gemm_variant != 0 || a.N < 192 || a.M % a.S || !(a.taps & 1) ||
                   in_dtype != out_dtype))
        return FS2_ERR_SHAPE;  // the gated store lives in the slab kernel's plain epilogue only
    const int variant = g_gemm_variant;  // 0 = auto, 1 = 128x128 register-staged, 2 = 128x256 DMA ring,
                                         // 3/4/5 = slab kernel with 128/192/256-row tiles
    const bool slab_ok = a.M % a.S == 0 && (a.taps & 1);
    const int ksp = a.ksplit > 1 ? a.ksplit : 1;
    if (a.w_presplit && (!(a.split || g_split_f32) || variant != 0 || !slab_ok || a.N < 192 || in_dtype != FS2_F32 || a.ksplit > 1 || a.zero_rows))
        return FS2_ERR_SHAPE;  // weights packed as heads + tails are the slab kernel's split arithmetic's own format
    if (a.C_lo && (fused || a.gate || a.stats_out || a.epi_res || a.zero_rows || a.ksplit > 1 || a.drop_p > 0.f || variant != 0 || !slab_ok ||
                   a.N < 192 || in_dtype != FS2_F32 || out_dtype != FS2_F32))
        return FS2_ERR_SHAPE;  // the head + tail store lives in the slab kernel's plain fp32 epilogue only
    if (ksp > 1 && (fused || a.bias || a.relu || a.gate || a.stats_out || a.epi_res || a.zero_rows || out_dtype != FS2_F32 || !slab_ok ||
                    a.N < 192 || (a.Cin / ke) % ksp || (variant != 0 && (variant < 3 || variant > 7))))
        return FS2_ERR_SHAPE;  // split-K: the slab kernel's plain fp32 store only
    if (variant >= 6 && variant <= 7 && slab_ok) {  // 6/7 = slab kernel with 32/64-row tiles
        if (fused) *fused = true;
        if (variant == 6) return launch_slab<1>(a, in_dtype, out_dtype, stream);
        return launch_slab<2>(a, in_dtype, out_dtype, stream);
    }
    if (variant >= 3 && variant <= 5 && slab_ok && !(fused && variant == 5)) {
        if (fused) *fused = true;
        if (variant == 3) return launch_slab<4>(a, in_dtype, out_dtype, stream);
        if (variant == 4 || a.stats_out || a.epi_res) return launch_slab<6>(a, in_dtype, out_dtype, stream);
        return launch_slab<8>(a, in_dtype, out_dtype, stream);
    }
    if (fused && variant != 0) return FS2_OK;  // forced non-slab kernel: caller falls back
    if (variant == 2) {
        if (a.stats_out || a.epi_res) return FS2_ERR_SHAPE;
        if (in_dtype == FS2_F32 && out_dtype == FS2_F32) return launch_glds_t<float, float>(a, stream);
        if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_glds_t<bf16, bf16>(a, stream);
        if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_glds_t<bf16, float>(a, stream);
        return FS2_ERR_SHAPE;
    }
    if (variant == 0 && slab_ok && a.N >= 192) {
        // One 512-thread workgroup per CU: pick the row-tile height that minimises
        // (rounds over the 256 CUs) x (tile rows + ~40 rows' worth of prologue/epilogue), measured
        // on MI355X (tools/bench_ops.py); fall back to the flat 128x128 kernel when utterances are
        // so short that per-utterance tiles would be mostly padding.
        const int S = a.taps == 1 ? a.M : a.S, nutt = a.M / S, tn = (a.N + S_BN - 1) / S_BN;
        int best = 0;
        long best_cost = 0, best_rows = 0;
        static const int kHeights[5] = {1, 2, 4, 6, 8};  // x32 rows; 32/64-row tiles keep small-M launches
        for (int hi = 0; hi < 5; ++hi) {                  // (the encoder's) spread over all CUs
            const int mi = kHeights[hi];
            if (fused && mi > 6) continue;                                    // 256-row tiles spill with the fused LayerNorm epilogue
            if ((a.stats_out || a.epi_res) && mi > 6) continue;                // ... and with the deferred one (r04: 24 spills even without its in-epilogue residual path; selected nowhere)
            const long bm = mi * 32, tm = (S + bm - 1) / bm;
            const long tiles = (long)nutt * tm * tn * ksp;
            long cost = ((tiles + 255) / 256) * (bm + 40);
            // long reductions (K >= 4096: the data-gradient convs of the training step, K = taps * filter): every workgroup
            // streams the whole K x 256 weight panel out of L2, and tiles x panel bytes over the ~10 TB/s the L2s deliver
            // together becomes the bound before the CUs fill - in the same units (one row of MFMA work per K) that is ~1 per
            // tile.  C5 encoder conv1 dgrad (M = 2048, N = 1024, K = 36864): 256 x 32-row tiles 500 us -> 128 x 64-row tiles.
            if (a.K / ksp >= 4096) cost = cost > tiles ? cost : tiles;
            if (!best || cost < best_cost) { best = mi; best_cost = cost; best_rows = (long)nutt * tm * bm; }
        }
        if (best_rows <= 2L * a.M || a.C_lo || a.w_presplit || a.rs_stats) {  // (the head + tail store / pre-split weights / row-scaled product exist in this kernel only)
            // more tiles than CUs: one workgroup per CU walks them, the next tile's first operands under this tile's epilogue
            // (gemm_persist.hip; same tile height, same arithmetic per element - bit-identical)
            if (!fused && tn_.gemm_persist && gemm_persist_supported(a, in_dtype, out_dtype, best) && gemm_persist_pays(a, best))
                return launch_gemm_persist(a, best, stream);
            if (fused) *fused = true;
            if (best == 1) return launch_slab<1>(a, in_dtype, out_dtype, stream);
            if (best == 2) return launch_slab<2>(a, in_dtype, out_dtype, stream);
            if (best == 4) return launch_slab<4>(a, in_dtype, out_dtype, stream);
            if (best == 6) return launch_slab<6>(a, in_dtype, out_dtype, stream);
            return launch_slab<8>(a, in_dtype, out_dtype, stream);
        }
    }
    if (fused) return FS2_OK;  // not the slab kernel: caller falls back to GEMM + LayerNorm kernel
    if (a.stats_out || a.epi_res || a.gate || ksp > 1 || a.drop_p > 0.f || a.rs_stats) return FS2_ERR_SHAPE;  // the deferred-LayerNorm epilogue lives in the slab kernel only
    if (in_dtype == FS2_F32 && out_dtype == FS2_F32) return launch_t<float, float>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_t<bf16, bf16>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_t<bf16, float>(a, stream);
    return FS2_ERR_SHAPE;
}

bool gemm_presplit_eligible(int N, int K) { return N >= 192 && K % 32 == 0; }

// fp32 (N, K) weights -> the split arithmetic's operand format, same size, in place or into `out`: per 32-channel chunk (128 bytes)
// 16-byte slot s < 4 = the bf16 heads of channels [4s, 4s + 4) and [16 + 4s, 16 + 4s + 4), slot 4 + s = their tails
__global__ __launch_bounds__(256) void presplit_pack_kernel(const float* w, unsigned char* out, size_t npairs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const size_t chunk = i >> 2;
    const int sl = (int)(i & 3);
    const uint4 c0 = *(const uint4*)(w + chunk * 32 + 4 * sl), c1 = *(const uint4*)(w + chunk * 32 + 16 + 4 * sl);
    uint4 h, l;
    split_bf16x3(c0, c1, h, l);
    *(uint4*)(out + chunk * 128 + sl * 16) = h;        // (in place: a thread writes the two slots it has read)
    *(uint4*)(out + chunk * 128 + (4 + sl) * 16) = l;
}
int launch_presplit_pack(const float* w, void* out, size_t n_values, hipStream_t stream) {
    if (n_values % 32) return FS2_ERR_SHAPE;
    if (!n_values) return FS2_OK;
    const size_t np = n_values / 8;
    hipLaunchKernelGGL(presplit_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, w, (unsigned char*)out, np);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// ---- split-K: planes -> output ----------------------------------------------------------------------------------------------
// out[i] = [out[i] +] ((p0 + p1) + p2) + ... in plane order (deterministic), 4 elements per thread, 16-byte plane loads
template <typename OutT>
__global__ __launch_bounds__(256) void split_k_reduce_kernel(const float* __restrict__ part, OutT* __restrict__ out, size_t n, int ksplit,
                                                             int accumulate) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a = *(const float4*)(part + i);
    for (int s = 1; s < ksplit; ++s) {
        const float4 b = *(const float4*)(part + (size_t)s * n + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if constexpr (sizeof(OutT) == 4) {
        float4* o = (float4*)(out + i);
        if (accumulate) { const float4 c = *o; a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; }
        *o = a;
    } else {
        uint2* o = (uint2*)(out + i);
        if (accumulate) {
            const uint2 c = *o;
            a.x += __uint_as_float(c.x << 16); a.y += __uint_as_float(c.x & 0xffff0000u);
            a.z += __uint_as_float(c.y << 16); a.w += __uint_as_float(c.y & 0xffff0000u);
        }
        *o = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
    }
}

int launch_split_k_reduce(const float* part, void* out, size_t n, int ksplit, int accumulate, int out_dtype, hipStream_t stream) {
    if (!part || !out || ksplit < 1 || n % 4 || (out_dtype != FS2_F32 && out_dtype != FS2_BF16)) return FS2_ERR_ARG;
    if (!n) return FS2_OK;
    const dim3 g((unsigned)((n / 4 + 255) / 256));
    if (out_dtype == FS2_F32) hipLaunchKernelGGL(split_k_reduce_kernel<float>, g, dim3(256), 0, stream, part, (float*)out, n, ksplit, accumulate);
    else hipLaunchKernelGGL(split_k_reduce_kernel<bf16>, g, dim3(256), 0, stream, part, (bf16*)out, n, ksplit, accumulate);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// A long reduction over few row tiles (the encoder-side data-gradient convs of the training step: M = 8192, N = 256, K = 9 x 1024)
// leaves every workgroup streaming the whole K x 256 weight panel at the CU's ~64 GB/s ingest rate with half the CUs idle
// (130 us at C2).  Split over K so that 128-row tiles fill the chip once: 1/ksplit of the panel per workgroup.
int gemm_splitk_choice(int M, int N, int Cin, int taps, int S, int in_dtype) {
    const int ke = in_dtype == FS2_BF16 ? 64 : 32;
    if (taps < 1 || !(taps & 1) || N < 192 || Cin % ke || (long)taps * Cin < 2048) return 1;
    if (taps == 1) S = M;
    if (S <= 0 || M % S) return 1;
    const long tiles = (long)(M / S) * ((S + 127) / 128) * ((N + S_BN - 1) / S_BN);
    int best = 1;
    for (int k = 2; k <= 8; k *= 2)
        if ((Cin / ke) % k == 0 && tiles * k <= 256 && (long)taps * Cin / k >= 1024) best = k;
    return best;
}

}  // namespace fs2

#ifdef FS2_SLAB_PROBE
extern "C" int fs2_dbg_slab_phase_stamps(unsigned long long* out /*4 x 8, host*/) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs2::g_slab_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
#endif
