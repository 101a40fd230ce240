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
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

static constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB: bytes of K per LDS row
static constexpr int TILE_BYTES = BM * ROWB;           // 16 KiB per operand per buffer

__device__ inline int swz(int row, int slot) { return row * ROWB + ((slot ^ (row & 7)) << 4); }

template <typename T, typename OutT>
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
static constexpr int G2_BM = 128, G2_BN = 256, G2_STAGES = 3;
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

template <typename T, typename OutT>
static int launch_t(const GemmArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    const size_t smem = 4 * TILE_BYTES;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_conv_kernel<T, OutT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return FS2_ERR_HIP;
        attr_set = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_conv_kernel<T, OutT>), dim3(tiles), dim3(256), smem, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

int g_gemm_variant = 0;

int launch_gemm(const GemmArgs& a, int in_dtype, int out_dtype, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0) return FS2_OK;
    const int ke = in_dtype == FS2_BF16 ? 64 : 32;
    const int e16 = in_dtype == FS2_BF16 ? 8 : 4;
    if (a.K % ke || a.Cin % ke || a.ldx % e16 || a.K != a.taps * a.Cin) return FS2_ERR_SHAPE;
    if (a.ldc % 4) return FS2_ERR_SHAPE;
    // big problems: the DMA-pipelined 128x256 kernel (needs enough tiles to cover the 256 CUs)
    const long big_tiles = (long)((a.M + G2_BM - 1) / G2_BM) * ((a.N + G2_BN - 1) / G2_BN);
    const int variant = g_gemm_variant;  // 0 = auto, 1 = force 128x128 register-staged, 2 = force DMA
    if (variant == 2 || (variant == 0 && a.N >= 192 && big_tiles >= 192)) {
        if (in_dtype == FS2_F32 && out_dtype == FS2_F32) return launch_glds_t<float, float>(a, stream);
        if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_glds_t<bf16, bf16>(a, stream);
        if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_glds_t<bf16, float>(a, stream);
        return FS2_ERR_SHAPE;
    }
    if (in_dtype == FS2_F32 && out_dtype == FS2_F32) return launch_t<float, float>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_t<bf16, bf16>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_t<bf16, float>(a, stream);
    return FS2_ERR_SHAPE;
}

}  // namespace fs2
