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
// (32 ch) with the SAME per-wave work; the slab is capped (voc_pick_mi16) so that two workgroups share a CU
// where the layer allows.  The narrow stages' resblocks run through vocoder_resblock.hip instead.
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

namespace {

template <typename T> struct VocT;
template <> struct VocT<bf16> { static constexpr int KE = 32; };   // elements per 64-byte k-step
template <> struct VocT<float> { static constexpr int KE = 16; };

// 0 < slope <= 1.  The bare instruction: fmaxf puts a canonicalising v_max(v, v) in front of every value that did not
// come out of an arithmetic instruction (vocoder_resblock.hip)
__device__ inline float lrelu(float v, float slope) {
    float r;
    const float m = v * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}

}  // namespace

// CP = padded input channels as a compile-time constant (0 = read it from the arguments): row stride,
// fragment offsets and swizzle fold into immediates; with a runtime stride every fragment address cost a
// VALU add and the K loop was issue-bound
template <typename T, int MI16, int CP>
__global__ __launch_bounds__(512) void vocoder_conv_kernel(VocConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slab[];
    constexpr int KE = VocT<T>::KE, RW = MI16 * 16;
    constexpr int E16 = 16 / (int)sizeof(T);  // elements per 16-byte piece

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WN = p.wn, WM = 8 / WN;
    const int wn = wave % WN, wm = wave / WN;
    const int fr = lane & 15, fg = lane >> 4;
    const int BM = WM * RW, halo = (p.taps - 1 + (p.shift_from ? 1 : 0)) * p.dil;
    const int tiles = (p.S + BM - 1) / BM;
    const int ub = blockIdx.x / tiles, tm = blockIdx.x % tiles, nt = blockIdx.y;
    const int len = p.lengths ? p.lengths[ub] * p.len_scale : p.S;  // valid rows of this utterance
    const int t0 = tm * BM;
    if (t0 >= len) return;  // block-uniform

    // ---- operand slab: index i <-> t = t0 - pad + i; 16-byte slot s of row i stored at SlabSwizzle::slot(s, i) ----
    const int cin_pad = CP ? CP : p.cin_pad;
    const int rowb = cin_pad * (int)sizeof(T), ns = rowb >> 4;  // slots per row (power of two, >= 4)
    const SlabSwizzle swz(ns);
    {
        const int rows = BM + halo, pieces = rows * ns, ns_sh = __builtin_ctz(ns);
        const size_t ubase = (size_t)ub * p.S;
        if (p.in_fp32) {  // conv_pre only (fp32 mel, 80 channels padded to cin_pad): 0.1 % of a pass
            for (int q = tid; q < pieces; q += 512) {
                const int i = q >> ns_sh, s = q & (ns - 1), t = t0 - p.pad + i, c = s * E16;
                float f[E16];
#pragma unroll
                for (int e = 0; e < E16; ++e) f[e] = 0.f;
                if (t >= 0 && t < len && c < p.cin) {
                    const float* src = (const float*)p.x + (ubase + t) * p.cin + c;
#pragma unroll
                    for (int e = 0; e < E16; e += 4) {
                        if (c + e < p.cin) {  // cin is a multiple of 4
                            const float4 v = *(const float4*)(src + e);
                            f[e] = v.x; f[e + 1] = v.y; f[e + 2] = v.z; f[e + 3] = v.w;
                        }
                    }
                }
                if (p.in_slope != 1.f) {
#pragma unroll
                    for (int e = 0; e < E16; ++e) f[e] = lrelu(f[e], p.in_slope);
                }
                *(uint4*)(slab + i * rowb + (swz.slot(s, i) << 4)) = Vec16<T>::pack(f);
            }
        } else {
            const T* xb = (const T*)p.x + ubase * p.cin;
            if (p.in_slope == 1.f && p.cin == cin_pad) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
                // nothing to apply while staging (the producer stored the activated values): straight global -> LDS
                // DMA, 1 KiB per wave instruction; lane l of chunk k lands at physical piece k*64 + l, so it fetches
                // the logical slot the swizzle keeps there; rows outside the utterance are out of the buffer's range
                // -> zeros (the launcher rounds the slab up to whole chunks)
                const __amdgpu_buffer_rsrc_t xrs =
                    __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)((size_t)len * rowb), 0x00020000);
                for (int k = wave; k * 64 < pieces; k += 8) {
                    const int P = k * 64 + lane, i = P >> ns_sh, ps = P & (ns - 1), t = t0 - p.pad + i;
                    const unsigned voff = (P < pieces && t >= 0 && t < len) ? (unsigned)(t * rowb + (swz.logical(ps, i) << 4)) : 0xFFFFF000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(slab + k * 1024), 16, voff,
                                                             0, 0, 0);
                }
                dma_drain();
#endif
            } else {
            // FB pieces per thread per trip, every load unconditional (clamped address, zeroed afterwards):
            // a trip costs ONE memory round trip, not one per piece behind its bounds branch
            constexpr int FB = 8;
            for (int q0 = tid; q0 < pieces; q0 += 512 * FB) {
                uint4 raw[FB];
                int dst[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int q = q0 + u * 512, qq = q < pieces ? q : pieces - 1;
                    const int i = qq >> ns_sh, s = qq & (ns - 1), t = t0 - p.pad + i, c = s * E16;
                    const bool ok = t >= 0 && t < len && c < p.cin;
                    const int tc = t < 0 ? 0 : (t < len ? t : len - 1), cc = c < p.cin ? c : 0;
                    dst[u] = q < pieces ? i * rowb + (swz.slot(s, i) << 4) : -1;
                    raw[u] = *(const uint4*)(xb + (size_t)tc * p.cin + cc);
                    if (!ok) raw[u] = make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    if (dst[u] < 0) continue;
                    if (p.in_slope != 1.f) {
                        float f[E16];
                        Vec16<T>::unpack(raw[u], f);
#pragma unroll
                        for (int e = 0; e < E16; ++e) f[e] = lrelu(f[e], p.in_slope);
                        raw[u] = Vec16<T>::pack(f);
                    }
                    *(uint4*)(slab + dst[u]) = raw[u];
                }
            }
            }
        }
    }

    // ---- weight stream: [n-tile][step][wave column][fragment][lane] x 16 B, 4-deep ring ----
    const int nkc = cin_pad / KE, nsteps = p.taps * nkc, nsteps4 = (nsteps + 3) & ~3;
    // (buffer loads: one per-lane byte offset for the whole launch, the step's offset in a scalar register)
    typedef unsigned vc_u4 __attribute__((ext_vector_type(4)));
    const int ntiles_w = p.post ? 1 : (p.n + WN * 32 - 1) / (WN * 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((size_t)ntiles_w * nsteps4 * WN * 2048), 0x00020000);
#endif
    const int wvoff = (wn * 128 + lane) * 16, wsbase = nt * nsteps4 * WN * 2048;
    auto loadB = [&](uint4 (&b)[2], int g) {
        g = g < nsteps4 ? g : nsteps4 - 1;
#if defined(__HIP_DEVICE_COMPILE__)
        const int so = wsbase + g * (WN * 2048);
        const vc_u4 v0 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, so, 0);
        const vc_u4 v1 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + 1024, so, 0);
        b[0] = make_uint4(v0.x, v0.y, v0.z, v0.w);
        b[1] = make_uint4(v1.x, v1.y, v1.z, v1.w);
#else
        (void)b;
#endif
    };
    uint4 bw[4][2];
    loadB(bw[0], 0);
    loadB(bw[1], 1);
    loadB(bw[2], 2);

    // the bias rides in as the accumulators' initial value (lane: channels n0 .. n0+7 of every row): the zero
    // fill costs the same moves and the epilogue loses an add per element (its VALU instructions and the MFMA
    // passes add up on a SIMD: this kernel's pipe is 67 % busy with another 32 % of VALU issue beside it)
    const int n0 = nt * WN * 32 + wn * 32 + fg * 8;
    f32x4_t acc[2][MI16];
    {
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (!p.post && n0 < p.n) {
            b0 = *(const float4*)(p.bias + n0);
            b1 = *(const float4*)(p.bias + n0 + 4);
        }
#pragma unroll
        for (int b = 0; b < MI16; ++b) {
            acc[0][b] = (f32x4_t){b0.x, b0.y, b0.z, b0.w};
            acc[1][b] = (f32x4_t){b1.x, b1.y, b1.z, b1.w};
        }
    }

    __syncthreads();

    const int wrow0 = wm * RW;
    const int nkc_shift = __builtin_ctz(nkc);
    const int tsh = (p.shift_from && nt * WN * 32 + wn * 32 >= p.shift_from) ? 1 : 0;  // wave-uniform
    // operand fragments run one group (PF fragments) ahead of the MFMAs that use them (two register sets, issue order
    // pinned): left to itself the compiler reads a fragment right before its MFMA and every MFMA group then
    // waits out an LDS round trip
    // this lane's fragment-0 address of step g0 + u of a trip (g0 % 4 == 0; u = 4: the next trip's first step).  With the
    // channel count a template parameter a step's channel block is a compile-time number after unrolling and the tap a scalar
    // that moves once per tap - decomposing a running step index cost ~19 scalar + ~10 vector instructions per step beside
    // its 16 MFMAs (vocoder_resblock.hip)
    constexpr int NKC = CP ? CP / KE : 0;
    auto a_addr = [&](const int g0, const int u) {
        int tap, kc;
        if constexpr (NKC >= 4) {        // a trip lies inside one tap
            tap = g0 / NKC;
            kc = (g0 & (NKC - 1)) + u;   // (g0 & (NKC - 1)) is 0 for NKC = 4
            if (kc >= NKC) { kc -= NKC; ++tap; }
        } else if constexpr (NKC > 0) {  // 4 / NKC taps per trip
            tap = g0 / NKC + u / NKC;
            kc = u % NKC;
        } else {
            tap = (g0 + u) >> nkc_shift;
            kc = (g0 + u) & (nkc - 1);
        }
        tap = tap < p.taps ? tap : p.taps - 1;  // padded steps multiply zero weights by any valid rows
        const int i0 = wrow0 + fr + (tap + tsh) * p.dil;
        return slab + i0 * rowb + (swz.slot(kc * 4 + fg, i0) << 4);
    };
    constexpr int PF = 2, NG = MI16 / PF;  // fragments per group, groups per step
    uint4 fx[2][PF];
    {
        const unsigned char* a0 = a_addr(0, 0);
#pragma unroll
        for (int mi = 0; mi < PF; ++mi) fx[0][mi] = *(const uint4*)(a0 + mi * 16 * rowb);
    }
#pragma unroll 1
    for (int g0 = 0; g0 < nsteps4; g0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u;
            loadB(bw[(u + 3) & 3], g + 3);
            const unsigned char* acur = a_addr(g0, u);
            const unsigned char* anext = a_addr(g0, u + 1);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int cur = (u * NG + q) & 1;  // 4 * NG groups per trip: the parity restarts at 0
                const unsigned char* na = q + 1 < NG ? acur + (q + 1) * PF * 16 * rowb : anext;
#pragma unroll
                for (int mi = 0; mi < PF; ++mi) fx[cur ^ 1][mi] = *(const uint4*)(na + mi * 16 * rowb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < PF; ++mi) Mma16<T>::step(bw[u][ni], fx[cur][mi], acc[ni][q * PF + mi]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: lane = rows (m*16 + fr), 8 consecutive channels n0 .. n0+7 ----
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
    // rows are addressed as a workgroup-uniform 64-bit base + a 32-bit per-lane byte offset (one utterance < 4 GiB)
    const size_t ubytes = (size_t)ub * p.S * p.n * sizeof(T);
    const char* resb = (const char*)p.res + ubytes;
    char* outb = (char*)p.out + ubytes;
    const unsigned rown = (unsigned)p.n * (unsigned)sizeof(T), nb0 = (unsigned)n0 * (unsigned)sizeof(T);
    // residual / previous-output rows are fetched MC rows at a time BEFORE any of them is used or stored
    // (unconditional loads from clamped rows): one memory round trip per chunk instead of two per row
    constexpr int NP = 8 / E16;  // 16-byte pieces of a lane's 8 channels
    constexpr int MC = MI16 == 14 ? (sizeof(T) == 2 ? 7 : 2) : (MI16 >= 4 ? 4 : 2);
#pragma unroll
    for (int m0 = 0; m0 < MI16; m0 += MC) {
        uint4 rr[MC][NP], oo[MC][NP];
        if (p.res) {
#pragma unroll
            for (int mm = 0; mm < MC; ++mm) {
                const int t = t0 + wrow0 + (m0 + mm) * 16 + fr, tc = t < len ? t : len - 1;
                const uint4* src = (const uint4*)(resb + ((unsigned)tc * rown + nb0));
#pragma unroll
                for (int q = 0; q < NP; ++q) rr[mm][q] = src[q];
            }
        }
        if (p.accumulate) {
#pragma unroll
            for (int mm = 0; mm < MC; ++mm) {
                const int t = t0 + wrow0 + (m0 + mm) * 16 + fr, tc = t < len ? t : len - 1;
                const uint4* src = (const uint4*)(outb + ((unsigned)tc * rown + nb0));
#pragma unroll
                for (int q = 0; q < NP; ++q) oo[mm][q] = src[q];
            }
        }
#pragma unroll
        for (int mm = 0; mm < MC; ++mm) {
            const int m = m0 + mm, t = t0 + wrow0 + m * 16 + fr;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = acc[r >> 2][m][r & 3];
            if (p.res) {
                float rv[8];
#pragma unroll
                for (int q = 0; q < NP; ++q) Vec16<T>::unpack(rr[mm][q], rv + q * E16);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += rv[r];
            }
            if (p.scale != 1.f) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] *= p.scale;
            }
            if (p.accumulate) {
                float ov[8];
#pragma unroll
                for (int q = 0; q < NP; ++q) Vec16<T>::unpack(oo[mm][q], ov + q * E16);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += ov[r];
            }
            if (p.out_slope != 1.f) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = lrelu(v[r], p.out_slope);
            }
            if (t < len) {
                uint4* dst = (uint4*)(outb + ((unsigned)t * rown + nb0));
#pragma unroll
                for (int q = 0; q < NP; ++q) dst[q] = Vec16<T>::pack(v + q * E16);
            }
        }
    }
}

// (Tuning::voc_lds_limit: KiB; 0 = heuristic - fs2_op_set_vocoder_lds_limit)

// rows per wave (x16) the slab of this layer leaves room for
static int voc_pick_mi16(const VocConvArgs& a, int esz, size_t* smem) {
    static const int cand[4] = {14, 8, 4, 2};
    const int WM = 8 / a.wn;
    // One maximal slab per CU leaves a workgroup alone with its fill -> multiply -> store chain; a cap
    // makes room for a second workgroup that covers it.  Measured on the V1 generator (32 x 1536
    // frames, every conv through this kernel): 62.6 ms per pass at 150 KiB, 46.8 at 76, 47.7 at 52, 51.2
    // at 36; with the narrow stages on the resident-resblock kernel: 42.1 / 39.0 / 39.8 / 40.6 ms at
    // 150 / 110 / 76 / 52 (110 = 128-row tiles at 256 channels, 256-row tiles x 2 workgroups at 128).
    const int g_voc_lds_limit = tuning_of(a.tune).voc_lds_limit;
    const size_t limit = (size_t)(g_voc_lds_limit > 0 ? g_voc_lds_limit : 110) * 1024;
    for (int c = 0; c < 4; ++c) {
        const size_t b = ((size_t)(WM * cand[c] * 16 + (a.taps - 1 + (a.shift_from ? 1 : 0)) * a.dil) * a.cin_pad * esz + 1023) & ~(size_t)1023;  // whole DMA chunks
        if (b <= limit || (c == 3 && b <= 150 * 1024)) {
            *smem = b;
            return cand[c];
        }
    }
    return 0;
}

template <typename T, int MI16, int CP>
static int voc_launch_c(const VocConvArgs& a, size_t smem, hipStream_t stream) {
    static size_t attr = 0;
    if (smem > attr) {
        if (hipFuncSetAttribute((const void*)vocoder_conv_kernel<T, MI16, CP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess)
            return FS2_ERR_HIP;
        attr = 150 * 1024;
    }
    const int BM = (8 / a.wn) * MI16 * 16;
    const int tiles = (a.S + BM - 1) / BM;
    const int ntiles = a.post ? 1 : (a.n + a.wn * 32 - 1) / (a.wn * 32);
    hipLaunchKernelGGL((vocoder_conv_kernel<T, MI16, CP>), dim3((unsigned)(tiles * a.B), (unsigned)ntiles), dim3(512), smem,
                       stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

template <typename T, int MI16>
static int voc_launch_t(const VocConvArgs& a, size_t smem, hipStream_t stream) {
    switch (a.cin_pad) {  // the V1 / V2 / V3 generators' widths; anything else takes the runtime-stride build
        case 32: return voc_launch_c<T, MI16, 32>(a, smem, stream);
        case 64: return voc_launch_c<T, MI16, 64>(a, smem, stream);
        case 128: return voc_launch_c<T, MI16, 128>(a, smem, stream);
        case 256: return voc_launch_c<T, MI16, 256>(a, smem, stream);
        case 512: return voc_launch_c<T, MI16, 512>(a, smem, stream);
        default: return voc_launch_c<T, MI16, 0>(a, smem, stream);
    }
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
    if (a.shift_from % 32 || a.shift_from < 0 || a.shift_from >= (a.post ? 1 : a.n)) return FS2_ERR_SHAPE;
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
