// Generic Conv1d for the HiFi-GAN generator (reference: litfass/third_party/hifigan/models.py:20-165):
//   out[t, n] = epilogue( sum_{tap, c} act(x[t + (tap - c0) * dil, c]) * W[n, tap, c] + bias[n] )
// time-major activations (rows = samples, channels contiguous), per-utterance zero "same" padding at
// EVERY layer (the reference synthesises one unpadded utterance at a time, __init__.py:37-43), input
// LeakyReLU applied while the operand is staged, epilogue = (+ residual) * scale (+ previous output)
// or tanh for conv_post.  ConvTranspose1d(k = 2s, stride s, pad s/2) is run through the same kernel
// as a 3-tap conv to s * Cout "phase-major" channels (see vocoder_engine.hip), because
// (T, s * Cout) row-major IS (T * s, Cout) row-major.
//
// Structure (the single-launch predictor's, predictor_fused.hip, generalised): 512 threads = 8 waves
// as WM x WN; every wave owns RW = 16 * MI16 rows x 32 output channels.  The workgroup keeps the
// whole (WM * RW + (k-1) * dil)-row x Cin operand slab in LDS (staged through registers: bounds,
// fp32->bf16, LeakyReLU), the weights stream from L2 straight into MFMA fragments (packed in
// fragment order at finalize; 4-deep register ring) and the K loop has no barrier.  Channel counts
// halve per stage while the sample count grows, so WM x WN goes 1x8 (256 ch) -> 2x4 -> 4x2 -> 8x1
// (32 ch) with the SAME per-wave work and an almost constant 112 KiB slab.
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

namespace {

template <typename T> struct VocT;
template <> struct VocT<bf16> { static constexpr int KE = 32; };   // elements per 64-byte k-step
template <> struct VocT<float> { static constexpr int KE = 16; };

__device__ inline float lrelu(float v, float slope) { return fmaxf(v, v * slope); }  // 0 < slope <= 1

}  // namespace

template <typename T, int MI16>
__global__ __launch_bounds__(512) void vocoder_conv_kernel(VocConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slab[];
    constexpr int KE = VocT<T>::KE, HF = MI16 / 2, RW = MI16 * 16;
    constexpr int E16 = 16 / (int)sizeof(T);  // elements per 16-byte piece

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WN = p.wn, WM = 8 / WN;
    const int wn = wave % WN, wm = wave / WN;
    const int fr = lane & 15, fg = lane >> 4;
    const int BM = WM * RW, halo = (p.taps - 1) * p.dil;
    const int tiles = (p.S + BM - 1) / BM;
    const int ub = blockIdx.x / tiles, tm = blockIdx.x % tiles, nt = blockIdx.y;
    const int len = p.lengths ? p.lengths[ub] * p.len_scale : p.S;  // valid rows of this utterance
    const int t0 = tm * BM;
    if (t0 >= len) return;  // block-uniform

    // ---- operand slab: index i <-> t = t0 - pad + i; 16-byte slot s of row i stored at s ^ swz(i) ----
    const int rowb = p.cin_pad * (int)sizeof(T), ns = rowb >> 4;            // slots per row (power of two, >= 4)
    const int sh = ns >= 16 ? 0 : (ns == 8 ? 1 : 2), smask = (ns >= 16 ? 16 : ns) - 1;
    {
        const int rows = BM + halo, pieces = rows * ns;
        const size_t ubase = (size_t)ub * p.S;
        // FB pieces per thread per trip: all their global loads are issued before the first is used
        constexpr int FB = 4;
        for (int q0 = tid; q0 < pieces; q0 += 512 * FB) {
            float f[FB][E16];
            int dst[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const int q = q0 + u * 512;
                const int i = q / ns, s = q - i * ns, t = t0 - p.pad + i, c = s * E16;
                dst[u] = q < pieces ? i * rowb + ((s ^ ((i >> sh) & smask)) << 4) : -1;
#pragma unroll
                for (int e = 0; e < E16; ++e) f[u][e] = 0.f;
                if (q < pieces && t >= 0 && t < len && c < p.cin) {
                    if (p.in_fp32) {
                        const float* src = (const float*)p.x + (ubase + t) * p.cin + c;
#pragma unroll
                        for (int e = 0; e < E16; e += 4) {
                            if (c + e < p.cin) {  // cin is a multiple of 4
                                const float4 v = *(const float4*)(src + e);
                                f[u][e] = v.x; f[u][e + 1] = v.y; f[u][e + 2] = v.z; f[u][e + 3] = v.w;
                            }
                        }
                    } else {
                        const uint4 v = *(const uint4*)((const T*)p.x + (ubase + t) * p.cin + c);
                        Vec16<T>::unpack(v, f[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                if (dst[u] < 0) continue;
                if (p.in_slope != 1.f) {
#pragma unroll
                    for (int e = 0; e < E16; ++e) f[u][e] = lrelu(f[u][e], p.in_slope);
                }
                *(uint4*)(slab + dst[u]) = Vec16<T>::pack(f[u]);
            }
        }
    }

    // ---- weight stream: [n-tile][step][wave column][fragment][lane] x 16 B, 4-deep ring ----
    const int nkc = p.cin_pad / KE, nsteps = p.taps * nkc, nsteps4 = (nsteps + 3) & ~3;
    const uint4* __restrict__ wbase = (const uint4*)p.w + ((size_t)nt * nsteps4 * WN + wn) * 128 + lane;
    const size_t wstep = (size_t)WN * 128;
    auto loadB = [&](uint4 (&b)[2], int g) {
        g = g < nsteps4 ? g : nsteps4 - 1;
        b[0] = wbase[g * wstep];
        b[1] = wbase[g * wstep + 64];
    };
    uint4 bw[4][2];
    loadB(bw[0], 0);
    loadB(bw[1], 1);
    loadB(bw[2], 2);

    f32x4_t acc[2][MI16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MI16; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    __syncthreads();

    const int wrow0 = wm * RW;
    const int nkc_shift = __builtin_ctz(nkc);
#pragma unroll 1
    for (int g0 = 0; g0 < nsteps4; g0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u;
            loadB(bw[(u + 3) & 3], g + 3);
            int tap = g >> nkc_shift;
            const int kc = g & (nkc - 1);
            tap = tap < p.taps ? tap : p.taps - 1;  // padded steps multiply zero weights by any valid rows
            const int i0 = wrow0 + fr + tap * p.dil;
            const unsigned char* arow_p = slab + i0 * rowb;
            const int acx = ((kc * 4 + fg) ^ ((i0 >> sh) & smask)) << 4;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                uint4 fx[HF];
#pragma unroll
                for (int mi = 0; mi < HF; ++mi) fx[mi] = *(const uint4*)(arow_p + (hf * HF + mi) * 16 * rowb + acx);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < HF; ++mi) Mma16<T>::step(bw[u][ni], fx[mi], acc[ni][hf * HF + mi]);
            }
        }
    }

    // ---- epilogue: lane = rows (m*16 + fr), 8 consecutive channels n0 .. n0+7 ----
    const int n0 = nt * WN * 32 + wn * 32 + fg * 8;
    if (p.post) {
        // conv_post + tanh (models.py:160-162): a single output channel, fp32 samples
        if (n0 == 0) {
            const float b0 = p.bias[0];
            float* out = (float*)p.out + (size_t)ub * p.S;
#pragma unroll
            for (int m = 0; m < MI16; ++m) {
                const int t = t0 + wrow0 + m * 16 + fr;
                if (t < len) out[t] = tanhf(acc[0][m][0] + b0);
            }
        }
        return;
    }
    if (n0 >= p.n) return;
    float bb[8];
    {
        const float4 b0 = *(const float4*)(p.bias + n0), b1 = *(const float4*)(p.bias + n0 + 4);
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    const size_t obase = (size_t)ub * p.S;
#pragma unroll
    for (int m = 0; m < MI16; ++m) {
        const int t = t0 + wrow0 + m * 16 + fr;
        if (t >= len) continue;
        const size_t o = (obase + t) * p.n + n0;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = acc[r >> 2][m][r & 3] + bb[r];
        if (p.res) {
            float rv[8];
            if constexpr (sizeof(T) == 2) {
                Vec16<T>::unpack(*(const uint4*)((const T*)p.res + o), rv);
            } else {
                Vec16<T>::unpack(*(const uint4*)((const T*)p.res + o), rv);
                Vec16<T>::unpack(*(const uint4*)((const T*)p.res + o + 4), rv + 4);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += rv[r];
        }
        if (p.scale != 1.f) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= p.scale;
        }
        T* dst = (T*)p.out + o;
        if (p.accumulate) {
            float ov[8];
            if constexpr (sizeof(T) == 2) {
                Vec16<T>::unpack(*(const uint4*)dst, ov);
            } else {
                Vec16<T>::unpack(*(const uint4*)dst, ov);
                Vec16<T>::unpack(*(const uint4*)(dst + 4), ov + 4);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += ov[r];
        }
        if constexpr (sizeof(T) == 2) {
            *(uint4*)dst = Vec16<T>::pack(v);
        } else {
            *(uint4*)dst = Vec16<T>::pack(v);
            *(uint4*)(dst + 4) = Vec16<T>::pack(v + 4);
        }
    }
}

int g_voc_lds_limit = 0;  // KiB; 0 = heuristic (tuning knob, fs2_op_set_vocoder_lds_limit)

// rows per wave (x16) the slab of this layer leaves room for
static int voc_pick_mi16(const VocConvArgs& a, int esz, size_t* smem) {
    static const int cand[4] = {14, 8, 4, 2};
    const int WM = 8 / a.wn;
    // One maximal slab per CU leaves a workgroup alone with its fill -> multiply -> store chain; a cap
    // makes room for a second workgroup that covers it.  Measured on the V1 generator (32 x 1536
    // frames, every conv through this kernel): 62.6 ms per pass at 150 KiB, 46.8 at 76, 47.7 at 52, 51.2
    // at 36; with the narrow stages on the resident-resblock kernel: 42.1 / 39.0 / 39.8 / 40.6 ms at
    // 150 / 110 / 76 / 52 (110 = 128-row tiles at 256 channels, 256-row tiles x 2 workgroups at 128).
    const size_t limit = (size_t)(g_voc_lds_limit > 0 ? g_voc_lds_limit : 110) * 1024;
    for (int c = 0; c < 4; ++c) {
        const size_t b = (size_t)(WM * cand[c] * 16 + (a.taps - 1) * a.dil) * a.cin_pad * esz;
        if (b <= limit || (c == 3 && b <= 150 * 1024)) {
            *smem = b;
            return cand[c];
        }
    }
    return 0;
}

template <typename T, int MI16>
static int voc_launch_t(const VocConvArgs& a, size_t smem, hipStream_t stream) {
    static size_t attr = 0;
    if (smem > attr) {
        if (hipFuncSetAttribute((const void*)vocoder_conv_kernel<T, MI16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess)
            return FS2_ERR_HIP;
        attr = 150 * 1024;
    }
    const int BM = (8 / a.wn) * MI16 * 16;
    const int tiles = (a.S + BM - 1) / BM;
    const int ntiles = a.post ? 1 : (a.n + a.wn * 32 - 1) / (a.wn * 32);
    hipLaunchKernelGGL((vocoder_conv_kernel<T, MI16>), dim3((unsigned)(tiles * a.B), (unsigned)ntiles), dim3(512), smem,
                       stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

int voc_steps_padded(int taps, int cin_pad, int dtype) {
    const int ke = dtype == FS2_BF16 ? 32 : 16;
    return (taps * (cin_pad / ke) + 3) & ~3;
}

int launch_vocoder_conv(const VocConvArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.S <= 0) return FS2_OK;
    if (a.wn != 1 && a.wn != 2 && a.wn != 4 && a.wn != 8) return FS2_ERR_SHAPE;
    if (a.cin_pad < 32 || (a.cin_pad & (a.cin_pad - 1)) || a.cin > a.cin_pad || a.cin % 4) return FS2_ERR_SHAPE;
    if (!a.post && (a.n % 32 || a.n % (a.wn * 32))) return FS2_ERR_SHAPE;
    if (!a.in_fp32 && a.cin % (dtype == FS2_BF16 ? 8 : 4)) return FS2_ERR_SHAPE;
    size_t smem = 0;
    const int mi = voc_pick_mi16(a, dtype == FS2_BF16 ? 2 : 4, &smem);
    if (!mi) return FS2_ERR_SHAPE;
    if (dtype == FS2_BF16) {
        if (mi == 14) return voc_launch_t<bf16, 14>(a, smem, stream);
        if (mi == 8) return voc_launch_t<bf16, 8>(a, smem, stream);
        if (mi == 4) return voc_launch_t<bf16, 4>(a, smem, stream);
        return voc_launch_t<bf16, 2>(a, smem, stream);
    }
    if (mi == 14) return voc_launch_t<float, 14>(a, smem, stream);
    if (mi == 8) return voc_launch_t<float, 8>(a, smem, stream);
    if (mi == 4) return voc_launch_t<float, 4>(a, smem, stream);
    return voc_launch_t<float, 2>(a, smem, stream);
}

}  // namespace fs2
