// HBM-bound row kernels of the FastSpeech2 forward for gfx950: everything that is not a GEMM or
// the attention core.  Activations are (B*S, C) row-major; one wave (64 lanes) owns one row
// wherever a row reduction is needed, loads are 8/16-byte vectors along the channel axis.
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

// ---- 4 consecutive channels per lane ---------------------------------------------------------
template <typename T> __device__ inline void load4(const T* p, float* f);
template <> __device__ inline void load4<float>(const float* p, float* f) {
    const float4 v = *(const float4*)p;
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ inline void load4<bf16>(const bf16* p, float* f) {
    const uint2 v = *(const uint2*)p;
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ inline void store4(T* p, const float* f);
template <> __device__ inline void store4<float>(float* p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ inline void store4<bf16>(bf16* p, const float* f) {
    *(uint2*)p = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
}

// =============================================================================================
// LayerNorm(x [+ res]) * gamma + beta, eps inside the sqrt — nn.LayerNorm as used by
// ConformerEncoderLayer.norm1/norm2 (model.py:114-115) and VarianceConvolutionLayer (model.py:538).
// Optional fused head: pred = masked_fill(Linear(filter,1)(y), mask, 0)  (model.py:512-518).
// =============================================================================================
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const T* x = (const T*)p.x + (size_t)row * p.H;
    const T* res = p.res ? (const T*)p.res + (size_t)row * p.H : nullptr;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.H) {
            load4<T>(x + c, v[i]);
            if (res) {
                float r[4];
                load4<T>(res + c, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] += r[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    float mean, rstd;
    if (p.pre_stats) {  // statistics left by the producing GEMM's epilogue: normalise only
        const float2* ps = (const float2*)p.pre_stats + (size_t)row * p.pre_parts;
        float2 pq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pq[q] = q < p.pre_parts ? ps[q] : make_float2(0.f, 0.f);
        const float s1 = (pq[0].x + pq[1].x) + (pq[2].x + pq[3].x), s2 = (pq[0].y + pq[1].y) + (pq[2].y + pq[3].y);
        mean = s1 / (float)p.H;
        rstd = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 / (float)p.H), 0.f) + p.eps);
    } else {
    mean = wave_sum(s) / (float)p.H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.H + p.eps);
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.H) {
            float g[4], bb[4], y[4];
            load4<float>(p.gamma + c, g);
            load4<float>(p.beta + c, bb);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            if (p.drop_p > 0.f) {  // the layer's nn.Dropout behind the LayerNorm (model.py:539,557), same mask as the stand-alone pass
                const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
                const float sc = 1.f / (1.f - p.drop_p);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = dropout_bits(p.drop_seed, p.drop_key, (uint64_t)row * p.H + c + e) >= thr ? y[e] * sc : 0.f;
            }
            if (p.y) store4<T>((T*)p.y + (size_t)row * p.H + c, y);
            if (p.dot_w) {
                float w[4];
                load4<float>(p.dot_w + c, w);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot += y[e] * w[e];
            }
        }
    }
    if (p.dot_w) {
        dot = wave_sum(dot) + (p.dot_b_dev ? *p.dot_b_dev : p.dot_b);
        if (lane == 0) p.pred[row] = (p.mask && p.mask[row]) ? 0.f : dot;
    }
}

int launch_layernorm(const LayerNormArgs& a, int dtype, hipStream_t stream) {
    if (a.M <= 0) return FS2_OK;
    if (a.H % 4 || a.H > 1024) return FS2_ERR_SHAPE;
    const int nv = (a.H + 255) / 256;
    const dim3 grid((a.M + 3) / 4), block(256);
#define FS2_LN(NVV)                                                                                     \
    if (nv == NVV) {                                                                                    \
        if (dtype == FS2_BF16) hipLaunchKernelGGL((layernorm_kernel<bf16, NVV>), grid, block, 0, stream, a); \
        else hipLaunchKernelGGL((layernorm_kernel<float, NVV>), grid, block, 0, stream, a);             \
    }
    FS2_LN(1) FS2_LN(2) FS2_LN(3) FS2_LN(4)
#undef FS2_LN
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// Row statistics of a pre-norm tensor, finished: the deferred-LayerNorm GEMM epilogue leaves one (sum, sum of squares) per row per
// 256-column tile; the row-scaled GEMM epilogue (GemmArgs::rs_stats) wants (rstd, rstd * mean) per row.  One thread per row.
__global__ __launch_bounds__(256) void rowstats_finish_kernel(const float2* __restrict__ parts, int nparts, float invn, float eps, float2* __restrict__ out, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float2 pq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pq[q] = q < nparts ? parts[(size_t)m * nparts + q] : make_float2(0.f, 0.f);
    const float s1 = (pq[0].x + pq[1].x) + (pq[2].x + pq[3].x), s2 = (pq[0].y + pq[1].y) + (pq[2].y + pq[3].y);  // the consumers' order of the parts
    const float mean = s1 * invn;
    const float rstd = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 * invn), 0.f) + eps);
    out[m] = make_float2(rstd, mean * rstd);
}
int launch_rowstats_finish(const float* parts, int nparts, int ncols, float eps, float* out, int M, hipStream_t stream) {
    if (M <= 0) return FS2_OK;
    if (!parts || !out || nparts < 1 || nparts > 4 || ncols <= 0) return FS2_ERR_ARG;
    hipLaunchKernelGGL(rowstats_finish_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, (const float2*)parts, nparts, 1.0f / (float)ncols, eps, (float2*)out, M);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// The Linear(N, 1) head behind a deferred LayerNorm, from what the GEMM epilogue left (GemmArgs::head_out): per row the statistic
// parts and the parts of sum_n v[n] * gw[n], gw = gamma * w.  pred = mask ? 0 : rstd * (dot - mean * sum_gw) + (beta . w + b)
// = LayerNorm(v) . w + b  (VariancePredictor's LayerNorm -> Linear -> masked_fill, model.py:512-518,538).  One thread per row.
__global__ __launch_bounds__(256) void head_finish_kernel(const float2* __restrict__ parts, const float* __restrict__ dots, int nparts, float invn, float eps,
                                                          float sum_gw, float cst, const uint8_t* __restrict__ mask, float* __restrict__ pred, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float2 pq[4];
    float dq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        pq[q] = q < nparts ? parts[(size_t)m * nparts + q] : make_float2(0.f, 0.f);
        dq[q] = q < nparts ? dots[(size_t)m * nparts + q] : 0.f;
    }
    const float s1 = (pq[0].x + pq[1].x) + (pq[2].x + pq[3].x), s2 = (pq[0].y + pq[1].y) + (pq[2].y + pq[3].y);
    const float dot = (dq[0] + dq[1]) + (dq[2] + dq[3]);
    const float mean = s1 * invn;
    const float rstd = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 * invn), 0.f) + eps);
    const float v = __builtin_fmaf(rstd, __builtin_fmaf(-mean, sum_gw, dot), cst);
    pred[m] = (mask && mask[m]) ? 0.f : v;
}
int launch_head_finish(const float* parts, const float* dots, int nparts, int ncols, float eps, float sum_gw, float cst, const uint8_t* mask, float* pred,
                       int M, hipStream_t stream) {
    if (M <= 0) return FS2_OK;
    if (!parts || !dots || !pred || nparts < 1 || nparts > 4 || ncols <= 0) return FS2_ERR_ARG;
    hipLaunchKernelGGL(head_finish_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, (const float2*)parts, dots, nparts, 1.0f / (float)ncols, eps, sum_gw, cst,
                       mask, pred, M);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =============================================================================================
// Depth-wise Conv1d over time, zero "same" padding per utterance, unmasked — conv1.0 of the
// LightSpeech FFN (model.py:75-81) and module.0 of the depth-wise predictor layer
// (model.py:545-551).  LDS-staged (TR + k - 1) x 64 slab, one output channel per lane.
// =============================================================================================
#ifndef FS2_DW_PAD
#define FS2_DW_PAD 0
#endif
static constexpr int DW_CT = 64, DW_KMAX = 32, DW_KP = 36, DW_LS = DW_CT + FS2_DW_PAD;  // DW_LS: slab row stride in LDS (tile rows: the kernel's TRT)
typedef float dw_f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ inline void load4v(const T* p, dw_f2& a, dw_f2& b);
template <> __device__ inline void load4v<float>(const float* p, dw_f2& a, dw_f2& b) {
    const float4 v = *(const float4*)p;
    a = dw_f2{v.x, v.y}; b = dw_f2{v.z, v.w};
}
template <> __device__ inline void load4v<bf16>(const bf16* p, dw_f2& a, dw_f2& b) {
    const uint2 v = *(const uint2*)p;
    a = dw_f2{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
    b = dw_f2{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
}

// Workgroup = 256 rows x 64 channels of one utterance; a thread owns 16 consecutive rows x 4 channels.  The
// (256 + k' - 1) x 64 input slab (k' = k rounded up to a multiple of the tap chunk TC; zeros outside the utterance) sits in LDS in
// the activation dtype, the k' x 64 weights (zeros past k) as fp32.  (r03: with 8 rows per thread and an fp32 slab the kernel moved
// 1.44 LDS bytes per multiply-add - 368 B of slab + weight reads per 256 - against a CU's 128 B and 128 multiply-adds per
// cycle: LDS-bound at 42 % of the HBM rate; 16 rows on a bf16 slab move 0.61.)  Taps go in chunks of TC: the chunk's TC x 4 weights
// sit in registers and the thread walks the 15 + TC input rows the chunk touches ONCE - one 8-byte LDS read feeds up
// to TC x 4 multiply-adds (input row o + j contributes tap c*TC + j to output row o).  Per output the taps are added in
// ascending order whatever TC is: the results do not depend on it, bit for bit.
// r05: the decoder-sized launches (k = 17 / 21 on 49152 rows) were bound by VALU issue, not by HBM (2.7 TB/s): (a) the chunk
// size follows k (launch_dwconv: the TC in {3, 5, 7, 8, 9} that pads k least - k = 17 computed 24 taps in chunks of 8, 18 in chunks
// of 9), (b) the multiply-adds go two channels at a time (v_pk_fma_f32 - the same fused multiply-add per channel).
// r06: the tile height is a template parameter (TRT = 256 or 128 rows, a thread owns TRT / 16 consecutive rows).  Measured
// (profiles/r06_v13_dwconv_tile_rows.txt): 256-row tiles win wherever they give every CU its three workgroups (all frame-level launches,
// C = 256 and 768: 12-18 / 29-52 us against 13-19 / 29-55 at 128 rows and 14-21 / 34-68 at 64), 128-row tiles where they do not (the
// encoder-side launches, 256 phonemes: C = 768 9.6-14.3 -> 8.3-12.8 us, C = 256 6.8-9.4 -> 5.4-6.8).  Per output the taps are added in
// the same order at either height: the choice is invisible in the results (test_dwconv_tile_heights_are_bit_identical).
template <typename T, int DW_TC, int TRT = 256>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? 3 : 1))) void dwconv_kernel(DwConvArgs p) {  // 3 workgroups per CU (LDS at 256 rows): <= 168 VGPRs
    constexpr int DW_TR = TRT, DW_RR = TRT / 16;
    __shared__ __attribute__((aligned(16))) T tile[(DW_TR + DW_KP - 1) * DW_LS];
    __shared__ __attribute__((aligned(16))) float wl[DW_KP * DW_CT];  // [tap][channel]
    __shared__ float rstat[(DW_TR + DW_KP - 1) * 2];                   // LayerNorm-on-load: (mean, rstd) per slab row
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * DW_TR, c0 = blockIdx.y * DW_CT, b = blockIdx.z;
    const T* x = (const T*)p.x + (size_t)b * p.S * p.C;
    const int nch = (p.k + DW_TC - 1) / DW_TC, kp = nch * DW_TC;
    const int rows = DW_TR + kp - 1;
    const bool full_c = c0 + DW_CT <= p.C;
    const bool lnl = p.ln_stats != nullptr;
    float lg[4] = {1.f, 1.f, 1.f, 1.f}, lb[4] = {0.f, 0.f, 0.f, 0.f};
    float2 pq[2][4];  // row-statistic parts of slab rows tid and tid + 256 (rows <= 291)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) pq[h][q] = make_float2(0.f, 0.f);
    if (lnl) {  // issued with the fill loads below
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = tid + 256 * h, t = t0 + r - p.pad;
            if (r < rows && t >= 0 && t < p.S) {
                const float2* ps = (const float2*)p.ln_stats + ((size_t)b * p.S + t) * p.ln_parts;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q < p.ln_parts) pq[h][q] = ps[q];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + (tid & 15) * 4 + e;
            if (c < p.C) { lg[e] = p.ln_g[c]; lb[e] = p.ln_b[c]; }
        }
    }
    // the taps of this channel tile, requested with the slab (r05: a load-then-store loop behind the fill cost a memory round trip per
    // 4 taps - the k-dependence of the launch time was this loop, not the multiply-adds)
    constexpr int WB = DW_KP * DW_CT / 256;
    float wv[WB];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
        const int i = tid + u * 256, tap = i / DW_CT, c = i % DW_CT;
        const bool ok = tap < p.k && c0 + c < p.C;
        wv[u] = p.w[ok ? (size_t)(c0 + c) * p.k + (p.flip ? p.k - 1 - tap : tap) : 0];
    }
    auto publish_rstat = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = tid + 256 * h;
            const float s1 = (pq[h][0].x + pq[h][1].x) + (pq[h][2].x + pq[h][3].x), s2 = (pq[h][0].y + pq[h][1].y) + (pq[h][2].y + pq[h][3].y);
            const float mean = s1 / (float)p.C;
            if (r < rows) {
                rstat[2 * r] = mean;
                rstat[2 * r + 1] = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 / (float)p.C), 0.f) + p.ln_eps);
            }
        }
        __syncthreads();
    };
    if (full_c) {
        // every load of the slab is issued before the first is used (unconditional, clamped row, zeroed afterwards):
        // ONE memory round trip for the fill instead of one per 256 pieces
        constexpr int FB = ((DW_TR + DW_KP - 1) * (DW_CT / 4) + 255) / 256;
        float v[FB][4];
        const int npc = rows * (DW_CT / 4);
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = tid + u * 256, ii = i < npc ? i : npc - 1;
            const int r = ii >> 4, cq = (ii & 15) * 4, t = t0 + r - p.pad;
            const int tc = t < 0 ? 0 : (t < p.S ? t : p.S - 1);
            load4<T>(x + (size_t)tc * p.C + c0 + cq, v[u]);
        }
        if (lnl) publish_rstat();
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = tid + u * 256, ii = i < npc ? i : npc - 1;
            const int r = ii >> 4, t = t0 + r - p.pad;
            if (lnl) {
                const float mean = rstat[2 * r], rstd = rstat[2 * r + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = __builtin_fmaf((v[u][e] - mean) * rstd, lg[e], lb[e]);
            }
            if (t < 0 || t >= p.S) v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = tid + u * 256;
            if (i < npc) store4<T>(tile + (i >> 4) * DW_LS + (i & 15) * 4, v[u]);
        }
    } else {
        if (lnl) publish_rstat();
        for (int i = tid; i < rows * (DW_CT / 4); i += 256) {  // channel tail tile: element by element
            const int r = i >> 4, cq = (i & 15) * 4;
            const int t = t0 + r - p.pad;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (t >= 0 && t < p.S) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + cq + e < p.C) {
                        v[e] = Num<T>::to_f32(x[(size_t)t * p.C + c0 + cq + e]);
                        if (lnl) v[e] = __builtin_fmaf((v[e] - rstat[2 * r]) * rstat[2 * r + 1], p.ln_g[c0 + cq + e], p.ln_b[c0 + cq + e]);
                    }
            }
            store4<T>(tile + r * DW_LS + cq, v);
        }
    }
#pragma unroll
    for (int u = 0; u < WB; ++u) {
        const int i = tid + u * 256, tap = i / DW_CT, c = i % DW_CT;
        if (i < DW_CT * kp) wl[i] = (tap < p.k && c0 + c < p.C) ? wv[u] : 0.f;
    }
    __syncthreads();
    const int cq = (tid & 15) * 4, r0 = (tid >> 4) * DW_RR;
    dw_f2 acc[DW_RR][2];
#pragma unroll
    for (int o = 0; o < DW_RR; ++o) acc[o][0] = acc[o][1] = dw_f2{0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        dw_f2 w[DW_TC][2];
#pragma unroll
        for (int j = 0; j < DW_TC; ++j) {
            const float4 wv = *(const float4*)(wl + (c * DW_TC + j) * DW_CT + cq);
            w[j][0] = dw_f2{wv.x, wv.y};
            w[j][1] = dw_f2{wv.z, wv.w};
        }
        const T* trow = tile + (r0 + c * DW_TC) * DW_LS + cq;
#pragma unroll
        for (int rr = 0; rr < DW_RR + DW_TC - 1; ++rr) {
            dw_f2 xa, xb;
            load4v<T>(trow + rr * DW_LS, xa, xb);
#pragma unroll
            for (int o = 0; o < DW_RR; ++o) {
                const int j = rr - o;  // compile-time after unrolling: tap c*TC + j of output row o
                if (j >= 0 && j < DW_TC) {
                    acc[o][0] = __builtin_elementwise_fma(w[j][0], xa, acc[o][0]);
                    acc[o][1] = __builtin_elementwise_fma(w[j][1], xb, acc[o][1]);
                }
            }
        }
    }
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[e] = (p.bias && c0 + cq + e < p.C) ? p.bias[c0 + cq + e] : 0.f;
    T* y = (T*)p.y + (size_t)b * p.S * p.C;
#pragma unroll
    for (int o = 0; o < DW_RR; ++o) {
        const int t = t0 + r0 + o;
        if (t >= p.S) break;
        float ov[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = acc[o][e >> 1][e & 1] + bias[e];
        if (full_c) {
            store4<T>(y + (size_t)t * p.C + c0 + cq, ov);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + cq + e < p.C) y[(size_t)t * p.C + c0 + cq + e] = Num<T>::from_f32(ov[e]);
        }
    }
}

int dwconv_tap_chunk(int k) {  // the chunk that pads k least; the larger one on a tie (fewer slab rows read twice)
    static const int cand[5] = {9, 8, 7, 5, 3};
    int best = 8, bestp = 1 << 30;
    for (int t : cand) {
        const int kp = (k + t - 1) / t * t;
        if (kp < bestp && kp <= DW_KP) { bestp = kp; best = t; }
    }
    return best;
}
int dwconv_tile_rows(int B, int S, int C) {  // 256 rows where that fills the chip's 3 x 256 workgroup slots once, else 128
    const long ct = (C + DW_CT - 1) / DW_CT;
    return (long)B * ((S + 255) / 256) * ct >= 768 ? 256 : 128;
}
int launch_dwconv(const DwConvArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.S <= 0) return FS2_OK;
    if (a.k < 1 || a.k > DW_KMAX || a.C % 4) return FS2_ERR_SHAPE;
    const int tr = dtype == FS2_BF16 ? dwconv_tile_rows(a.B, a.S, a.C) : 256;
    const dim3 grid((a.S + tr - 1) / tr, (a.C + DW_CT - 1) / DW_CT, a.B), block(256);
    switch (dwconv_tap_chunk(a.k)) {
#define FS2_DW(TC)                                                                                                     \
    case TC:                                                                                                           \
        if (dtype != FS2_BF16) hipLaunchKernelGGL((dwconv_kernel<float, TC>), grid, block, 0, stream, a);               \
        else if (tr == 256) hipLaunchKernelGGL((dwconv_kernel<bf16, TC, 256>), grid, block, 0, stream, a);              \
        else hipLaunchKernelGGL((dwconv_kernel<bf16, TC, 128>), grid, block, 0, stream, a);                             \
        break;
        FS2_DW(3) FS2_DW(5) FS2_DW(7) FS2_DW(8) FS2_DW(9)
#undef FS2_DW
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =============================================================================================
// x[b,l,:] = (E[phones[b,l]] + pe[l]) + spk[b]; src_mask = phones == 0
// (fastspeech2.py:651-660, model.py:53-55,143).  Row 0 of E is the zero padding row, PE and the
// speaker vector are still added at pad positions.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B * p.L) return;
    const int b = row / p.L, l = row % p.L;
    long long ph = p.phones[row];
    if (lane == 0) p.src_mask[row] = ph == 0;
    if (ph < 0 || ph >= p.n_phones) ph = 0;  // caller validates ids on the host side of the ABI
    const float* e = p.table + (size_t)ph * p.H;
    const float* pe = p.pe + (size_t)l * p.H;
    const float* sp = p.spk + (size_t)b * p.H;
    T* x = (T*)p.x + (size_t)row * p.H;
    for (int c = lane * 4; c < p.H; c += 256) {
        float a[4], q[4], s[4], o[4];
        load4<float>(e + c, a);
        load4<float>(pe + c, q);
        load4<float>(sp + c, s);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __fadd_rn(__fadd_rn(a[i], q[i]), s[i]);
        store4<T>(x + c, o);
    }
}

int launch_embed(const EmbedArgs& a, int dtype, hipStream_t stream) {
    if (a.B * a.L <= 0) return FS2_OK;
    if (a.H % 4) return FS2_ERR_SHAPE;
    const dim3 grid((a.B * a.L + 3) / 4), block(256);
    if (dtype == FS2_BF16) hipLaunchKernelGGL(embed_kernel<bf16>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(embed_kernel<float>, grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// spk[b,:] = relu(W dvec[b] + bias)  — SpeakerEmbedding.forward, model.py:137-143 (computed once
// per forward; the reference evaluates it twice, fastspeech2.py:658,707, with identical results)
__global__ __launch_bounds__(256) void spk_proj_kernel(SpkProjArgs p) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= p.B * p.H) return;
    const int b = o / p.H, n = o % p.H;
    const float* w = p.w + (size_t)n * p.Din;
    const float* d = p.dvec + (size_t)b * p.Din;
    float s = 0.f;
    for (int k = lane; k < p.Din; k += 64) s = fmaf(w[k], d[k], s);
    s = wave_sum(s) + p.b[n];
    if (lane == 0) p.spk[o] = fmaxf(s, 0.f);
}

int launch_spk_proj(const SpkProjArgs& a, hipStream_t stream) {
    if (a.B <= 0) return FS2_OK;
    hipLaunchKernelGGL(spk_proj_kernel, dim3((a.B * a.H + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// key-padding byte mask -> one 64-bit valid word per 64 keys (consumed by the attention kernel)
__global__ __launch_bounds__(64) void mask_bits_kernel(MaskBitsArgs p) {
    const int w = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int key = w * 64 + lane;
    const bool valid = key < p.S && p.mask[(size_t)b * p.S + key] == 0;
    const unsigned long long bal = __ballot(valid);
    if (lane == 0) p.bits[(size_t)b * p.nw64 + w] = bal;
}

int launch_mask_bits(const MaskBitsArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.nw64 <= 0) return FS2_OK;
    hipLaunchKernelGGL(mask_bits_kernel, dim3(a.nw64, a.B), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =============================================================================================
// Duration rounding + zero-duration guard + prefix sum (model.py:299-309 and the lengths the
// LengthRegulator derives, model.py:350-354).  One workgroup per utterance.
//   d = int(clamp(round_half_even(exp(p) - 1), 0));  if sum_valid d <= n_valid // 2: d[valid] = 1
// =============================================================================================
__device__ inline int dur_from_pred(float p) {
    const float v = rintf(__fsub_rn(expf(p), 1.0f));  // torch.round = half-to-even
    return v > 0.f ? (int)v : 0;
}

__global__ __launch_bounds__(256) void durations_kernel(DurationArgs p) {
    __shared__ int red[2][4];
    __shared__ int wsum[4];
    __shared__ int guard_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* dp = p.dur_pred + (size_t)b * p.L;
    const uint8_t* mk = p.src_mask + (size_t)b * p.L;
    const int32_t* forced = p.forced ? p.forced + (size_t)b * p.L : nullptr;
    int guard = 0;
    if (!forced) {
        int sd = 0, nv = 0;
        for (int l = tid; l < p.L; l += 256) {
            if (!mk[l]) { sd += dur_from_pred(dp[l]); nv += 1; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, 64); nv += __shfl_xor(nv, o, 64); }
        if (lane == 0) { red[0][wave] = sd; red[1][wave] = nv; }
        __syncthreads();
        if (tid == 0) {
            const int s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
            const int n = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            guard_s = s <= n / 2;
        }
        __syncthreads();
        guard = guard_s;
    }
    int running = 0;
    for (int base = 0; base < p.L; base += 256) {
        const int l = base + tid;
        int d = 0;
        if (l < p.L) {
            if (forced) d = forced[l];
            else d = (guard && !mk[l]) ? 1 : dur_from_pred(dp[l]);
            if (d < 0) d = 0;
        }
        int inc = d;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o, 64);
            if (lane >= o) inc += n;
        }
        __syncthreads();  // wsum reuse across iterations
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (l < p.L) {
            p.dur[(size_t)b * p.L + l] = d;
            p.cum[(size_t)b * p.L + l] = off + inc;
        }
        running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (tid == 0) {
        p.totals[b] = running;
        p.guard[b] = guard;
    }
}

int launch_durations(const DurationArgs& a, hipStream_t stream) {
    if (a.B <= 0) return FS2_OK;
    hipLaunchKernelGGL(durations_kernel, dim3(a.B), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =============================================================================================
// Length regulator as a prefix-sum gather (LengthRegulator.forward, model.py:349-370):
//   y[b,t,:] = x[b,p,:] with cum[p-1] <= t < cum[p]  for t < total_b, else 0;
//   tgt_mask[b,t] = t >= total_b  (the UNtruncated total: a clipped utterance has no pad).
// =============================================================================================
// One wave per RG_ROWS consecutive frames of one utterance.  The search "which phone owns frame t" is
// not a per-row chain of dependent loads: the wave keeps the utterance's prefix sums in registers (one
// coalesced load, RG_MAXC per lane) and the owner of t is the NUMBER of prefix sums <= t, a ballot +
// popcount per register - exact integer arithmetic, same result as the upper_bound it replaces.
constexpr int RG_ROWS = 16, RG_MAXC = 16;  // up to 64 * 16 = 1024 phones per utterance on the fast path
template <typename T>
__global__ __launch_bounds__(256) void regulate_kernel(RegulateArgs p) {
    const int lane = threadIdx.x & 63;
    const int tb = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RG_ROWS, b = blockIdx.y;
    if (tb >= p.T) return;
    const int total = p.totals[b];
    const int32_t* cum = p.cum + (size_t)b * p.L;
    const int nvec = p.H * (int)sizeof(T) / 16;
    const bool fast = p.L <= 64 * RG_MAXC;
    int32_t cl[RG_MAXC];
    if (fast) {
#pragma unroll
        for (int k = 0; k < RG_MAXC; ++k) {
            const int i = lane + 64 * k;
            cl[k] = i < p.L ? cum[i] : 0x7fffffff;
        }
    }
    const int nc = (p.L + 63) / 64;
    // rows narrower than a wave's 1 KiB (H = 256 bf16: 512 B) are copied two at a time, one per half wave
    const bool pair = nvec <= 32;
    const int half = lane >> 5, hl = lane & 31;
    for (int r = 0; r < RG_ROWS; r += pair ? 2 : 1) {
        if (tb + r >= p.T) break;
        int lo_of[2] = {0, 0};
        bool in_of[2] = {false, false};
#pragma unroll
        for (int q = 0; q < 2; ++q) {  // the owner search is a whole-wave ballot: done for both rows by all lanes
            const int t = tb + r + q;
            if (q == 1 && !pair) break;
            if (t >= p.T) break;
            in_of[q] = t < total;
            if (lane == 0) p.tgt_mask[(size_t)b * p.T + t] = t >= total;
            if (t < total) {
                int lo = 0;
                if (fast) {
#pragma unroll
                    for (int k = 0; k < RG_MAXC; ++k)
                        if (k < nc) lo += __popcll(__ballot(cl[k] <= t));
                } else {
                    int hi = p.L;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (cum[mid] > t) hi = mid; else lo = mid + 1;
                    }
                }
                lo_of[q] = lo;
            }
        }
        if (pair) {
            const int t = tb + r + half;
            if (t < p.T && hl < nvec) {
                uint4* dst = (uint4*)((T*)p.y + ((size_t)b * p.T + t) * p.H);
                const bool in = half ? in_of[1] : in_of[0];
                const int lo = half ? lo_of[1] : lo_of[0];
                dst[hl] = in ? ((const uint4*)((const T*)p.x + ((size_t)b * p.L + lo) * p.H))[hl] : make_uint4(0, 0, 0, 0);
            }
        } else {
            const int t = tb + r;
            uint4* dst = (uint4*)((T*)p.y + ((size_t)b * p.T + t) * p.H);
            if (in_of[0]) {
                const uint4* src = (const uint4*)((const T*)p.x + ((size_t)b * p.L + lo_of[0]) * p.H);
                for (int i = lane; i < nvec; i += 64) dst[i] = src[i];
            } else {
                for (int i = lane; i < nvec; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
            }
        }
    }
}

int launch_regulate(const RegulateArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.T <= 0) return FS2_OK;
    const int esz = dtype == FS2_BF16 ? 2 : 4;
    if ((a.H * esz) % 16) return FS2_ERR_SHAPE;
    const dim3 grid((a.T + 4 * RG_ROWS - 1) / (4 * RG_ROWS), a.B), block(256);
    if (dtype == FS2_BF16) hipLaunchKernelGGL(regulate_kernel<bf16>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(regulate_kernel<float>, grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =============================================================================================
// Variance embedding add (VarianceEncoder.forward inference branch, model.py:434-438, and the
// adaptor's x = x + emb, model.py:333):  idx = bucketize(pred*std + mean, bins) (right=False);
// y = x + Emb[idx]; optionally the decoder-side  y = (y + pe[t]) + spk[b]  (fastspeech2.py:705-718)
// fused behind it.  The compare is fp32 mul-then-add (no FMA) like the reference's tensor ops.
// =============================================================================================
// One wave per BE_ROWS consecutive rows.  bucketize = lower_bound over the sorted bin edges = the NUMBER
// of edges < v: the wave holds the edges in registers (BE_MAXB per lane, loaded once) and counts with
// a ballot + popcount per register instead of walking a chain of dependent loads per row.
constexpr int BE_ROWS = 8, BE_MAXB = 8;  // up to 512 edges on the fast path
template <typename T, int NV>
__global__ __launch_bounds__(256) void bucket_embed_kernel(BucketArgs p) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * BE_ROWS;
    const int M = p.B * p.T;
    if (row0 >= M) return;
    const int nedge = p.nbins - 1;
    const bool fast = nedge <= 64 * BE_MAXB;
    float bl[BE_MAXB];
    if (p.pred && fast) {
#pragma unroll
        for (int k = 0; k < BE_MAXB; ++k) {
            const int i = lane + 64 * k;
            bl[k] = i < nedge ? p.bins[i] : INFINITY;  // +inf < v is false for every v: never counted
        }
    }
    const int nb = (nedge + 63) / 64;
    for (int r = 0; r < BE_ROWS; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        const int b = row / p.T, t = row % p.T;
        const float* e = nullptr;
        if (p.pred) {
            const float src = p.bucket_src ? p.bucket_src[row] : p.pred[p.pred_per_utt ? b : row];
            const float v = __fadd_rn(__fmul_rn(src, p.std), p.mean);
            int lo = 0;
            if (fast) {
#pragma unroll
                for (int k = 0; k < BE_MAXB; ++k)
                    if (k < nb) lo += __popcll(__ballot(bl[k] < v));
            } else {
                int hi = nedge;  // lower_bound over nbins-1 boundaries
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (p.bins[mid] < v) lo = mid + 1; else hi = mid;
                }
            }
            if (p.forced_idx) {
                lo = p.forced_idx[row];
                lo = lo < 0 ? 0 : (lo >= p.nbins ? p.nbins - 1 : lo);
            }
            if (lane == 0 && p.idx_out) p.idx_out[row] = lo;
            e = p.emb + (size_t)lo * p.H;
        }
        const T* x = (const T*)p.x + (size_t)row * p.H;
        T* y = (T*)p.y + (size_t)row * p.H;
        const float* pe = p.pe ? p.pe + (size_t)t * p.H : nullptr;
        const float* sp = p.spk ? p.spk + (size_t)b * p.H : nullptr;
        if constexpr (NV > 0) {
            // H <= 1024: the row's NV 256-column chunks unrolled, every load of the row requested before the first add (r05: the
            // rolled loop was three dependent load -> add -> store trips per row)
            float v[NV][4], ae[NV][4], ap[NV][4], as[NV][4];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < p.H) {
                    load4<T>(x + c, v[i]);
                    if (e) load4<float>(e + c, ae[i]);
                    if (pe) load4<float>(pe + c, ap[i]);
                    if (sp) load4<float>(sp + c, as[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < p.H) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (e) v[i][k] = __fadd_rn(v[i][k], ae[i][k]);
                        if (pe) v[i][k] = __fadd_rn(v[i][k], ap[i][k]);
                        if (sp) v[i][k] = __fadd_rn(v[i][k], as[i][k]);
                    }
                    store4<T>(y + c, v[i]);
                }
            }
        } else
        for (int c = lane * 4; c < p.H; c += 256) {
            float v[4], a[4];
            load4<T>(x + c, v);
            if (e) {
                load4<float>(e + c, a);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], a[i]);
            }
            if (pe) {
                load4<float>(pe + c, a);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], a[i]);
            }
            if (sp) {
                load4<float>(sp + c, a);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], a[i]);
            }
            store4<T>(y + c, v);
        }
    }
}

int launch_bucket_embed(const BucketArgs& a, int dtype, hipStream_t stream) {
    if (a.B * a.T <= 0) return FS2_OK;
    if (a.H % 4) return FS2_ERR_SHAPE;
    const dim3 grid((a.B * a.T + 4 * BE_ROWS - 1) / (4 * BE_ROWS)), block(256);
    switch (a.H <= 1024 ? (a.H + 255) / 256 : 0) {
#define FS2_BE(NVV)                                                                                             \
    case NVV:                                                                                                   \
        if (dtype == FS2_BF16) hipLaunchKernelGGL((bucket_embed_kernel<bf16, NVV>), grid, block, 0, stream, a); \
        else hipLaunchKernelGGL((bucket_embed_kernel<float, NVV>), grid, block, 0, stream, a);                  \
        break;
        FS2_BE(0) FS2_BE(1) FS2_BE(2) FS2_BE(3) FS2_BE(4)
#undef FS2_BE
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// =============================================================================================
// CWT pitch head (model.py:412-431,445-461; dataset/cwt.py:18-21,49-50).  Two launches, one workgroup per utterance:
//   mean_std[b] = mean_std_linear(mean_t out_conv[b, t, :])            (time mean over ALL T rows, pads included)
//   s[b, t] = sum_j spec[b, t, j] (pads: 0);  pred = (s - mean_t s) / (std_t s + 1e-7) * std_b + mean_b  (unbiased std)
// =============================================================================================
__device__ inline float block_sum_256(float v, float* sh) {  // all 256 threads get the total; fixed order
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void cwt_mean_std_kernel(CwtArgs p) {
    __shared__ float sh[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const T* x = (const T*)p.out_conv + (size_t)b * p.T * p.F;
    float d0 = 0.f, d1 = 0.f;
    for (int c = tid; c < p.F; c += 256) {
        float s = 0.f;
        for (int t = 0; t < p.T; ++t) s += Num<T>::to_f32(x[(size_t)t * p.F + c]);
        const float m = s / (float)p.T;
        d0 = fmaf(m, p.ms_w[c], d0);
        d1 = fmaf(m, p.ms_w[p.F + c], d1);
    }
    d0 = block_sum_256(d0, sh);
    d1 = block_sum_256(d1, sh);
    if (tid == 0) {
        p.mean_std[2 * b] = d0 + p.ms_b[0];
        p.mean_std[2 * b + 1] = d1 + p.ms_b[1];
    }
}

__global__ __launch_bounds__(256) void cwt_recompose_kernel(CwtArgs p) {
    __shared__ float sh[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* pr = p.pred + (size_t)b * p.T;
    float acc = 0.f;
    for (int t = tid; t < p.T; t += 256) {
        const size_t row = (size_t)b * p.T + t;
        const bool pad = p.mask && p.mask[row];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const float v = pad ? 0.f : p.spec[row * p.ld_spec + j];
            if (p.spec_out) p.spec_out[row * 10 + j] = v;
            s += v;
        }
        pr[t] = s;
        acc += s;
    }
    const float m = block_sum_256(acc, sh) / (float)p.T;
    float q = 0.f;
    for (int t = tid; t < p.T; t += 256) { const float d = pr[t] - m; q = fmaf(d, d, q); }
    const float sd = sqrtf(block_sum_256(q, sh) / (float)(p.T - 1));  // torch.std: unbiased (T = 1 -> NaN, as the reference)
    const float mean_b = p.mean_std[2 * b], std_b = p.mean_std[2 * b + 1];
    for (int t = tid; t < p.T; t += 256) pr[t] = __fadd_rn(__fmul_rn((pr[t] - m) / (sd + 1e-7f), std_b), mean_b);
}

int launch_cwt_head(const CwtArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.T <= 0) return FS2_OK;
    if (dtype == FS2_BF16) hipLaunchKernelGGL(cwt_mean_std_kernel<bf16>, dim3(a.B), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(cwt_mean_std_kernel<float>, dim3(a.B), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(cwt_recompose_kernel, dim3(a.B), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
