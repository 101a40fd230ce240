// HiFi-GAN ResBlock "1" (reference: litfass/third_party/hifigan/models.py:98-104) on LDS-resident tiles:
//     for (c1, c2, d) in pairs:  x = c2(lrelu(c1_d(lrelu(x)))) + x
// one (c1, c2) pair per launch, or all three pairs of a block where the convs' K loops are short
// (voc_resblock_mi16), for the 32 / 64 / 128-channel stages.  Conv by conv these stages move ~2 GB of
// activations per conv for a few GFLOP each and sit on HBM / launch latency.  Here a
// workgroup keeps its tile of the residual stream AND of the intermediate in LDS (the single-launch
// predictor's scheme, predictor_fused.hip):
//   slab X = lrelu(x) of the tile (+ guard rows), slab Y = lrelu(c1 output); both hold the
//   ACTIVATED values because that is what the next conv multiplies; the raw residual is recovered
//   from X by the inverse map (a < 0 ? a / slope : a) in the epilogue of c2, which then overwrites
//   its own elements of X in place.  Rows outside the utterance are written as zeros after every
//   conv (the reference pads every conv of a single unpadded utterance with zeros).
//   Weights of the convs lie back to back in fragment order; one 4-deep register ring streams
//   them from L2 across conv boundaries; no barrier inside a K loop, two per conv pair.
// Every conv but the first (whose guard rows hold real samples) makes (k-1)/2 * dil more rows at both tile
// edges stale, so a tile of R rows finishes R - 2H rows, H = (k-1)/2 * (sum over pairs of (d + 1) - d_first)
// - for a single pair just the (k-1)/2 rows of its c2; tiles overlap by that halo.
#include <type_traits>

#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

namespace {
template <typename T> struct RbT;
template <> struct RbT<bf16> { static constexpr int KE = 32; };
template <> struct RbT<float> { static constexpr int KE = 16; };
// 0 < slope <= 1: LeakyReLU(v) = max(v, slope*v), its inverse = min(a, a/slope).  fmaxf / fminf put a
// canonicalising v_max(v, v) in front of every value that did not come out of an arithmetic instruction (MFMA
// results, unpacked bf16): 3 VALU instructions per element, and the epilogues of the narrow stages are VALU-bound
// (32 channels, k = 3: ~800 VALU instructions = 3200 cycles per conv beside 1024 cycles of MFMA).  The bare
// instruction returns the same value for every non-NaN input; the multiply pairs up into v_pk_mul_f32.
__device__ inline float rb_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline float rb_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// N values at a time, the multiplies on float pairs (v_pk_mul_f32)
typedef float rb_f2 __attribute__((ext_vector_type(2)));
template <int N> __device__ inline void rb_lrelu_n(float (&v)[N], float slope) {
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int r = 0; r < N; r += 2) {
        const rb_f2 x = {v[r], v[r + 1]};
        const rb_f2 m = x * slope;
        v[r] = rb_max(x.x, m.x);
        v[r + 1] = rb_max(x.y, m.y);
    }
}
__device__ inline void rb_lrelu8(float (&v)[8], float slope) { rb_lrelu_n<8>(v, slope); }
// v = acc + (a < 0 ? a / slope : a)
__device__ inline void rb_add_raw8(float (&v)[8], const float (&a)[8], float inv_slope) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const rb_f2 x = {a[r], a[r + 1]};
        const rb_f2 m = x * inv_slope;
        const rb_f2 raw = {rb_min(x.x, m.x), rb_min(x.y, m.y)};
        const rb_f2 acc = {v[r], v[r + 1]};
        const rb_f2 o = acc + raw;
        v[r] = o.x;
        v[r + 1] = o.y;
    }
}
}  // namespace

// CH = channels (32 / 64 / 128) as a compile-time constant: row strides, fragment offsets and the swizzle
// fold into immediates - with a runtime row stride every fragment address cost its own VALU add and the
// K loop was issue-bound (35 scalar/vector instructions per 8 MFMAs)
// NW = waves per workgroup: 8, or 4 for the half-height tile that shares a CU with a second workgroup - two
// INDEPENDENT 4-wave workgroups (one wave per SIMD each) instead of eight lock-step waves: the issue arbiter
// serves the oldest wave first, so of two waves of ONE workgroup on a SIMD the younger reaches every barrier
// ~40 % late (measured with s_memtime stamps) while the older one idles there
// MID = a pair in the middle of a block (out_act: scale 1, nothing to add to, result stored activated) - its
// store loop carries neither the scale multiply nor the previous-output path; !MID carries no output LeakyReLU
template <typename T, int MI16, int CH, int NW, bool MID>
__global__ __launch_bounds__(NW * 64, 2) void vocoder_resblock_kernel(VocResblockArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int KE = RbT<T>::KE, RW = MI16 * 16;
    constexpr int E16 = 16 / (int)sizeof(T), SPL = 8 / E16;  // slots per lane per row (8 channels)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int C = CH, WN = CH / 32, WM = NW / WN, NT = NW * 64;
    const int wn = wave % WN, wm = wave / WN;
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int R = WM * RW;
    const int c = (p.taps - 1) / 2;
    int dmax = 1, dsum = 0;
    for (int m = 0; m < p.npairs; ++m) {
        dmax = p.dil[m] > dmax ? p.dil[m] : dmax;
        dsum += p.dil[m] + 1;
    }
    // X guard: widest dilated conv; Y guard: c2 (dil 1).  The X guard rows are loaded with real samples, so the
    // very first conv loses no rows: a single pair only gives up the c rows of its c2 at each edge
    const int H = c * (dsum - p.dil[0]), G = c * dmax, GY = c, V = R - 2 * H;
    const int tiles = (p.S + V - 1) / V;
    const int ub = blockIdx.x / tiles, tm = blockIdx.x % tiles;
    const int len = p.lengths ? p.lengths[ub] * p.len_scale : p.S;
    const int t0 = tm * V;
    if (t0 >= len) return;  // block-uniform
    const int tbase = t0 - H;  // time of tile row 0; slab index i <-> tile row i - G

    constexpr int rowb = C * (int)sizeof(T), ns = rowb >> 4;
    const SlabSwizzle swz(ns);
    const int srows = R + 2 * G;
    unsigned char* slabX = lds;                           // slab index i <-> tile row i - G
    unsigned char* slabY = lds + (size_t)srows * rowb;    // slab index i <-> tile row i - GY
    auto slot_off = [&](int i, int s) { return i * rowb + (swz.slot(s, i) << 4); };

    // ---- X <- lrelu(x) for every slab row (guards included), zeros outside the utterance; Y guards <- 0
    {
        const T* xb = (const T*)p.x + (size_t)ub * p.S * C;
        const int pieces = srows * ns;
        constexpr int ns_sh = ns == 4 ? 2 : (ns == 8 ? 3 : (ns == 16 ? 4 : (ns == 32 ? 5 : 6)));
        if (p.x_act) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
            // the producer stored lrelu(x): straight global -> LDS DMA, 1 KiB (64 pieces) per wave instruction, no
            // VGPRs and no VALU; lane l of chunk k lands at physical piece k*64 + l, so it FETCHES the logical slot
            // that the swizzle keeps there; rows outside the utterance are out of the buffer's range -> zeros.
            // (A last partial chunk spills zeros into Y's leading guard rows, which are zero anyway.)
            const __amdgpu_buffer_rsrc_t xrs =
                __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)((size_t)len * rowb), 0x00020000);
            for (int k = wave; k * 64 < pieces; k += NW) {
                const int P = k * 64 + lane, i = P >> ns_sh, ps = P & (ns - 1), t = tbase - G + i;
                const unsigned voff = (P < pieces && t >= 0 && t < len) ? (unsigned)(t * rowb + (swz.logical(ps, i) << 4)) : 0xFFFFF000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(slabX + k * 1024), 16, voff, 0,
                                                         0, 0);
            }
#endif
        } else {
        // FB pieces per thread per trip, every load unconditional (clamped row, zeroed afterwards): one
        // memory round trip per trip
        constexpr int FB = 12;  // one trip covers a 4-wave tile
        for (int q0 = tid; q0 < pieces; q0 += NT * FB) {
            uint4 raw[FB];
            int dst[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const int q = q0 + u * NT, qq = q < pieces ? q : pieces - 1;
                const int i = qq >> ns_sh, s = qq & (ns - 1), t = tbase - G + i;
                const int tc = t < 0 ? 0 : (t < len ? t : len - 1);
                dst[u] = q < pieces ? slot_off(i, s) : -1;
                raw[u] = *(const uint4*)((const char*)xb + (unsigned)(tc * rowb + (s << 4)));  // uniform base + 32-bit offset
                if (t < 0 || t >= len) raw[u] = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                if (dst[u] < 0) continue;
                float f[E16];
                Vec16<T>::unpack(raw[u], f);
                rb_lrelu_n<E16>(f, p.slope);
                *(uint4*)(slabX + dst[u]) = Vec16<T>::pack(f);
            }
        }
        }
        const int gp = 2 * GY * ns;
        for (int q = tid; q < gp; q += NT) {
            int i = q / ns;
            const int s = q - i * ns;
            if (i >= GY) i += R;
            *(uint4*)(slabY + slot_off(i, s)) = make_uint4(0u, 0u, 0u, 0u);
        }
    }

    // ---- weight stream over all six convs: [conv][step][wave column][fragment][lane] x 16 B ----
    constexpr int nkc = C / KE;
    const int nsteps = p.taps * nkc, nsteps4 = (nsteps + 3) & ~3;
    const int nconv = 2 * p.npairs, total = nconv * nsteps4;
    // (buffer loads: one per-lane byte offset for the whole launch, the step's offset in a scalar register - the flat form
    // built a 64-bit address per step)
    typedef unsigned rb_u4 __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((size_t)total * WN * 2048), 0x00020000);
#endif
    const int wvoff = (wn * 128 + lane) * 16;
    auto loadB = [&](uint4 (&b)[2], int g) {
        g = g < total ? g : total - 1;
#if defined(__HIP_DEVICE_COMPILE__)
        const int so = g * (WN * 2048);
        const rb_u4 v0 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, so, 0);
        const rb_u4 v1 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + 1024, so, 0);
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

    const int wrow0 = wm * RW;
    constexpr int nkc_shift = nkc == 1 ? 0 : (nkc == 2 ? 1 : (nkc == 4 ? 2 : (nkc == 8 ? 3 : 4)));
    const int n0 = wn * 32 + fg * 8;  // this lane's 8 consecutive channels
    const float inv_slope = 1.0f / p.slope;
    // output rows are addressed as a workgroup-uniform 64-bit base + a 32-bit per-lane byte offset (an utterance
    // is < 4 GiB): per-row 64-bit pointer arithmetic was ~100 VALU instructions per conv
    char* ob = (char*)p.out + (size_t)ub * p.S * rowb;
    const unsigned nb = (unsigned)n0 * (unsigned)sizeof(T);
    if (p.x_act) dma_drain();  // this wave's slab DMAs have landed before the barrier publishes them
    __syncthreads();

    // byte offsets of this lane's 8 output channels in its fragment-0 row of X / Y; the swizzle reads row bits 0-2
    // only, so fragment m lies m * 16 rows further on at the same slot (an immediate offset)
    int xo[SPL], yo[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
        xo[q] = slot_off(G + wrow0 + fr, n0 / E16 + q);
        yo[q] = slot_off(GY + wrow0 + fr, n0 / E16 + q);
    }
    const bool edge = tbase + wrow0 < 0 || tbase + wrow0 + RW > len;  // this wave's rows cross an utterance end

    // one conv: SECOND = the c2 of a pair (reads Y, adds the residual, writes X or - the last one - the output)
    auto conv = [&](auto second_c, const int j) {
        constexpr bool second = decltype(second_c)::value;
        const int dil = second ? 1 : p.dil[j >> 1];
        const unsigned char* src = second ? slabY : slabX;
        const int ibase = (second ? GY : G) - c * dil + wrow0 + fr;  // slab index of this lane's fragment-0 row at tap 0
        const bool last = second && j == nconv - 1;

        // the block's result is added to the previous output: those rows are requested before the K loop that hides
        // them (unconditional loads, clamped rows)
        // (bf16; the fp32 build has no registers for that and fetches them chunk by chunk in the epilogue)
        constexpr bool OO_EARLY = sizeof(T) == 2;
        constexpr int MC = sizeof(T) == 2 ? MI16 : 2;  // rows of X (and of the previous output) in flight, x16
        uint4 oo[(!MID && second) ? (OO_EARLY ? MI16 : MC) : 1][SPL];
        if constexpr (!MID && second && OO_EARLY) {
            if (last && p.accumulate) {
#pragma unroll
                for (int m = 0; m < MI16; ++m) {
                    const int tt = tbase + wrow0 + m * 16 + fr, tc = tt < 0 ? 0 : (tt < len ? tt : len - 1);
                    const uint4* po = (const uint4*)(ob + (unsigned)(tc * rowb) + nb);
#pragma unroll
                    for (int q = 0; q < SPL; ++q) oo[m][q] = po[q];
                }
            }
        }

        // the bias is the C operand of every accumulator's first MFMA (lane: channels n0 .. n0+7 of every row): copying
        // it into the 16 accumulators first was 56-64 v_mov per conv
        f32x4_t acc[2][MI16], bia[2];
        {
            const float* bp = p.bias + j * C + n0;
            const float4 b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 4);
            bia[0] = (f32x4_t){b0.x, b0.y, b0.z, b0.w};
            bia[1] = (f32x4_t){b1.x, b1.y, b1.z, b1.w};
        }
        // operand fragments run one group (PF fragments) ahead of the MFMAs that use them (two register sets, issue order
        // pinned): left to itself the compiler reads a fragment right before its MFMA and every MFMA group
        // then waits out an LDS round trip
        // this lane's fragment-0 address of step g0 + u of a trip (g0 % 4 == 0; u may run into the next trip).  The channel
        // block kc of the step is a compile-time number after unrolling (nkc <= 4) and the tap a scalar that moves once per
        // 4 / nkc steps: decomposing a running step index per step was 7-19 scalar + ~10 vector instructions per step beside
        // its 16 MFMAs - more than their issue shadow holds
        auto a_addr = [&](const int g0, const int u) {
            int tap, kc;
            if constexpr (nkc <= 4) {
                tap = (g0 >> nkc_shift) + (u >> nkc_shift);
                kc = u & (nkc - 1);
            } else {
                tap = (g0 + u) >> nkc_shift;
                kc = (g0 + u) & (nkc - 1);
            }
            tap = tap < p.taps ? tap : p.taps - 1;  // padded steps: zero weights x any valid rows
            const int i0 = ibase + tap * dil;
            return src + i0 * rowb + (swz.slot(kc * 4 + fg, i0) << 4);
        };
        // PF fragments per group, NG groups per step, NS register sets: the set freed by group i-1 is refilled
        // with group i + NS - 1 while group i multiplies (4 * NG groups per trip, a multiple of NS)
        // (2 fragments x 4 sets measured 1.7 % faster per pass than 4 x 2 on the same registers; 4 x 4 and 2 x 8 no better)
        constexpr int PF = 2, NG = MI16 / PF, NS = 4, D = NS - 1, NA = (NG - 1 + D) / NG + 1;
        static_assert((4 * NG) % NS == 0, "set index must restart every trip");
        uint4 fx[NS][PF];
#pragma unroll
        for (int gi = 0; gi < D; ++gi) {
            const unsigned char* a0 = a_addr(0, gi / NG) + (gi % NG) * PF * 16 * rowb;
#pragma unroll
            for (int mi = 0; mi < PF; ++mi) fx[gi][mi] = *(const uint4*)(a0 + mi * 16 * rowb);
        }
        auto trip = [&](auto first_c, const int g0) {  // four K steps; the conv's first trip starts the accumulators
            constexpr bool first = decltype(first_c)::value;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u;
                loadB(bw[(u + 3) & 3], j * nsteps4 + g + 3);
                const unsigned char* as[NA];
#pragma unroll
                for (int k = 0; k < NA; ++k) as[k] = a_addr(g0, u + k);
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    const int gi = u * NG + q, cur = gi % NS, tgt = (gi + D) % NS;
                    const unsigned char* na = as[(q + D) / NG] + ((q + D) % NG) * PF * 16 * rowb;
#pragma unroll
                    for (int mi = 0; mi < PF; ++mi) fx[tgt][mi] = *(const uint4*)(na + mi * 16 * rowb);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int mi = 0; mi < PF; ++mi) {
                            if (first && u == 0) acc[ni][q * PF + mi] = bia[ni];
                            Mma16<T>::step(bw[u][ni], fx[cur][mi], acc[ni][q * PF + mi]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        trip(std::true_type{}, 0);
#pragma unroll 1
        for (int g0 = 4; g0 < nsteps4; g0 += 4) trip(std::false_type{}, g0);

        // ---- epilogue: lane = rows (m*16 + fr), channels n0 .. n0+7 ----
        if constexpr (!second) {  // Y <- lrelu(c1)
#pragma unroll
            for (int m = 0; m < MI16; ++m) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = acc[r >> 2][m][r & 3];
                rb_lrelu8(v, p.slope);
#pragma unroll
                for (int q = 0; q < SPL; ++q) *(uint4*)(slabY + yo[q] + m * 16 * rowb) = Vec16<T>::pack(v + q * E16);
            }
        } else {
            // + residual, recovered from the activated copy; all of this wave's rows of X are requested first
            uint4 xa[MC][SPL];
            auto fetch = [&](int m) {  // at the head of every chunk of MC fragments
#pragma unroll
                for (int mm = 0; mm < MC; ++mm)
#pragma unroll
                    for (int q = 0; q < SPL; ++q) xa[mm][q] = *(const uint4*)(slabX + xo[q] + (m + mm) * 16 * rowb);
            };
            auto with_res = [&](int m, float (&v)[8]) {
                float a[8];
#pragma unroll
                for (int q = 0; q < SPL; ++q) Vec16<T>::unpack(xa[m % MC][q], a + q * E16);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = acc[r >> 2][m][r & 3];
                rb_add_raw8(v, a, inv_slope);
            };
            if (!last) {  // X <- lrelu(x + c2), in place
#pragma unroll
                for (int m = 0; m < MI16; ++m) {
                    if (m % MC == 0) fetch(m);
                    float v[8];
                    with_res(m, v);
                    rb_lrelu8(v, p.slope);
#pragma unroll
                    for (int q = 0; q < SPL; ++q) *(uint4*)(slabX + xo[q] + m * 16 * rowb) = Vec16<T>::pack(v + q * E16);
                }
            } else {
#pragma unroll
                for (int m = 0; m < MI16; ++m) {
                    if (m % MC == 0) {
                        fetch(m);
                        if constexpr (!MID && !OO_EARLY) {
                            if (p.accumulate) {
#pragma unroll
                                for (int mm = 0; mm < MC; ++mm) {
                                    const int tt = tbase + wrow0 + (m + mm) * 16 + fr, tc = tt < 0 ? 0 : (tt < len ? tt : len - 1);
                                    const uint4* po = (const uint4*)(ob + (unsigned)(tc * rowb) + nb);
#pragma unroll
                                    for (int q = 0; q < SPL; ++q) oo[mm][q] = po[q];
                                }
                            }
                        }
                    }
                    float v[8];
                    with_res(m, v);
                    const int row = wrow0 + m * 16 + fr, t = tbase + row;
                    if constexpr (MID) {  // the next pair of this block reads it with x_act
                        rb_lrelu8(v, p.slope);
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= p.scale;
                        if (p.accumulate) {
                            float ov[8];
#pragma unroll
                            for (int q = 0; q < SPL; ++q) Vec16<T>::unpack(oo[OO_EARLY ? m : m % MC][q], ov + q * E16);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] += ov[r];
                        }
                    }
                    if (t >= 0 && t < len && row >= H && row < H + V) {
                        uint4* dst = (uint4*)(ob + (unsigned)(t * rowb) + nb);
#pragma unroll
                        for (int q = 0; q < SPL; ++q) dst[q] = Vec16<T>::pack(v + q * E16);
                    }
                }
            }
        }
        if (!last) {
            // rows outside the utterance are the next conv's zero padding: only a wave whose rows cross an utterance
            // end re-zeroes them (a per-element select in the loop above cost 64 VALU instructions per conv)
            if (edge) {
                unsigned char* dstb = second ? slabX : slabY;
#pragma unroll
                for (int m = 0; m < MI16; ++m) {
                    const int t = tbase + wrow0 + m * 16 + fr;
                    if (t < 0 || t >= len) {
#pragma unroll
                        for (int q = 0; q < SPL; ++q) *(uint4*)(dstb + (second ? xo[q] : yo[q]) + m * 16 * rowb) = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
            __syncthreads();
        }
    };

#pragma unroll 1
    for (int pr = 0; pr < p.npairs; ++pr) {
        conv(std::false_type{}, 2 * pr);
        conv(std::true_type{}, 2 * pr + 1);
    }
}

// (Tuning::voc_fused_resblock - fs2_op_set_vocoder_fused_resblock: 0 off, 1 auto, 2 pairs only, 4 = full-height 8-wave tiles only)

static void rb_geom(const VocResblockArgs& a, int nw, int mi16, int* R, int* H, int* G) {
    const int c = (a.taps - 1) / 2;
    int dmax = 1, dsum = 0;
    for (int m = 0; m < a.npairs; ++m) {
        dmax = a.dil[m] > dmax ? a.dil[m] : dmax;
        dsum += a.dil[m] + 1;
    }
    *R = (nw / a.wn) * mi16 * 16;
    *H = c * (dsum - a.dil[0]);
    *G = c * dmax;
}
static size_t rb_lds_bytes(const VocResblockArgs& a, int nw, int mi16, int esz) {
    int R, H, G;
    rb_geom(a, nw, mi16, &R, &H, &G);
    return (size_t)((R + 2 * G) + (R + 2 * ((a.taps - 1) / 2))) * a.C * esz;
}

// tile shape for this launch (npairs = 3: whole resblock, 1: one (c1, c2) pair) as 100 * waves + (16-row
// fragments per wave), 0 = not worth it / does not fit: the LDS must hold both slabs and the halo may eat at
// most a fifth of the tile (measured: a whole 64-channel k=11 block at 77 % useful rows is slower than conv by
// conv).  A 4-wave workgroup on <= 76 KiB (two per CU, covering each other's fills, barriers and epilogues)
// is taken when >= 85 % of its rows are useful: 3-9 % faster than the 8-wave full-height tile on the 32/64-
// channel stages and on the 128-channel k=7 pairs, equal elsewhere (repeated A/B runs).
int voc_resblock_mi16(const VocResblockArgs& a, int dtype) {
    const int g_voc_fused_resblock = tuning_of(a.tune).voc_fused_resblock;
    if (!g_voc_fused_resblock || (g_voc_fused_resblock == 2 && a.npairs != 1)) return 0;
    const int esz = dtype == FS2_BF16 ? 2 : 4;
    if (a.C != 32 && a.C != 64 && a.C != 128) return 0;
    if (a.wn != a.C / 32 || !(a.taps & 1) || (a.npairs != 1 && a.npairs != 3)) return 0;
    // a whole block per launch pays while a conv's K loop is short (<= 7 steps of 32 channels): measured 5-15 %
    // faster than three pair launches at 32 ch k=3/7 and 64 ch k=3, 5-7 % slower at 64 ch k=7 and 32 ch k=11
    if (a.npairs == 3 && a.taps * a.C > 224 && g_voc_fused_resblock != 7) return 0;
    const bool try_half = g_voc_fused_resblock != 4;
    if (try_half) {
        int R, H, G;
        rb_geom(a, 4, 8, &R, &H, &G);
        const int pct = g_voc_fused_resblock >= 50 ? g_voc_fused_resblock : 85;  // knob >= 50: threshold in percent (tuning)
        if (rb_lds_bytes(a, 4, 8, esz) <= 76 * 1024 && (R - 2 * H) * 100 >= R * pct) return 408;
    }
    static const int cand[2] = {8, 4};
    for (int k = 0; k < 2; ++k) {
        int R, H, G;
        rb_geom(a, 8, cand[k], &R, &H, &G);
        if (rb_lds_bytes(a, 8, cand[k], esz) <= 150 * 1024 && (R - 2 * H) * 5 >= R * 4) return 800 + cand[k];
    }
    return 0;
}

template <typename T, int MI16, int CH, int NW, bool MID>
static int rb_launch_c(const VocResblockArgs& a, size_t smem, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)vocoder_resblock_kernel<T, MI16, CH, NW, MID>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return FS2_ERR_HIP;
        attr = true;
    }
    int R, H, G;
    rb_geom(a, NW, MI16, &R, &H, &G);
    const int V = R - 2 * H;
    const int tiles = (a.S + V - 1) / V;
    hipLaunchKernelGGL((vocoder_resblock_kernel<T, MI16, CH, NW, MID>), dim3((unsigned)(tiles * a.B)), dim3(NW * 64), smem,
                       stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

template <typename T, int MI16, int NW>
static int rb_launch_t(const VocResblockArgs& a, size_t smem, hipStream_t stream) {
    const bool mid = a.out_act != 0;
    if (mid && (a.npairs != 1 || a.scale != 1.f || a.accumulate)) return FS2_ERR_ARG;
    if (a.C == 32) return mid ? rb_launch_c<T, MI16, 32, NW, true>(a, smem, stream) : rb_launch_c<T, MI16, 32, NW, false>(a, smem, stream);
    if (a.C == 64) return mid ? rb_launch_c<T, MI16, 64, NW, true>(a, smem, stream) : rb_launch_c<T, MI16, 64, NW, false>(a, smem, stream);
    return mid ? rb_launch_c<T, MI16, 128, NW, true>(a, smem, stream) : rb_launch_c<T, MI16, 128, NW, false>(a, smem, stream);
}

int launch_vocoder_resblock(const VocResblockArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.S <= 0) return FS2_OK;
    const int cfg = voc_resblock_mi16(a, dtype);
    if (!cfg) return FS2_ERR_SHAPE;
    const int nw = cfg / 100, mi = cfg % 100;
    const size_t smem = rb_lds_bytes(a, nw, mi, dtype == FS2_BF16 ? 2 : 4);
    if (dtype == FS2_BF16) {
        if (nw == 4) return rb_launch_t<bf16, 8, 4>(a, smem, stream);
        return mi == 8 ? rb_launch_t<bf16, 8, 8>(a, smem, stream) : rb_launch_t<bf16, 4, 8>(a, smem, stream);
    }
    if (nw == 4) return rb_launch_t<float, 8, 4>(a, smem, stream);
    return mi == 8 ? rb_launch_t<float, 8, 8>(a, smem, stream) : rb_launch_t<float, 4, 8>(a, smem, stream);
}

}  // namespace fs2
