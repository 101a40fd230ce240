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

int launch_gemm(const GemmArgs& a, int in_dtype, int out_dtype, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0) return FS2_OK;
    const int ke = in_dtype == FS2_BF16 ? 64 : 32;
    const int e16 = in_dtype == FS2_BF16 ? 8 : 4;
    if (a.K % ke || a.Cin % ke || a.ldx % e16 || a.K != a.taps * a.Cin) return FS2_ERR_SHAPE;
    if (a.ldc % 4) return FS2_ERR_SHAPE;
    if (in_dtype == FS2_F32 && out_dtype == FS2_F32) return launch_t<float, float>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_BF16) return launch_t<bf16, bf16>(a, stream);
    if (in_dtype == FS2_BF16 && out_dtype == FS2_F32) return launch_t<bf16, float>(a, stream);
    return FS2_ERR_SHAPE;
}

}  // namespace fs2
