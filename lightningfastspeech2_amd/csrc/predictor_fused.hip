// Whole VariancePredictor in one launch (reference: litfass/fastspeech2/model.py:482-522, dense
// VarianceConvolutionLayer :525-561): n x [Conv1d(256 -> 256, k=3, same) -> ReLU -> LayerNorm] ->
// Linear(256, 1) -> masked_fill, bf16 storage / fp32 arithmetic.
//
// Why a dedicated kernel: as five conv+LN launches every layer is one tile per CU with an exposed
// load -> 12 K-steps -> store chain (25-27 us each, ~30 % of the MFMA rate) and a 50 MB HBM round
// trip of activations nobody needs.  Here a workgroup keeps a (16*MI16 + 2)-row x 256-channel slab of
// ONE utterance in LDS for the whole predictor:
//   * the slab is filled once by buffer-load-to-LDS DMA (rows outside the utterance read zeros =
//     the conv's "same" padding);
//   * every layer's weights stream from L2 straight into MFMA B-fragments (pre-packed in fragment
//     order at fs2_finalize, 1 KiB contiguous per wave-load, 4-deep register ring that runs across
//     layer boundaries), so the K loop has NO barrier and no weight LDS traffic;
//   * the conv+ReLU+LN epilogue writes the next layer's input back into the slab in place (rows
//     outside [0, S) as zeros); only the last layer's scalar head leaves the chip.
// Each layer makes one more row at both slab edges stale (its neighbour was not recomputed), so a
// tile of R = 16*MI16 rows yields R - 2(n-1) finished rows; tiles overlap by that halo.
#include "fs2_common.h"
#include "fs2_kernels.h"
#include <type_traits>

namespace fs2 {

namespace {
constexpr int PF_H = 256, PF_TAPS = 3, PF_KB = PF_H / 32;  // a dense layer: 3 taps x 8 k-steps of 32
constexpr int PF_STEP_U4 = 8 * 2 * 64;  // 16-byte fragments per k-step: [wave][fragment][lane]
constexpr int PF_ROWB = PF_H * 2;       // slab row bytes (bf16)
#ifndef FS2_PF_PREFETCH
#define FS2_PF_PREFETCH 1
#endif
constexpr bool PF_PREFETCH = FS2_PF_PREFETCH;
// scheduling fence around the fragment reads of a half: hipcc otherwise sinks the reads to just before their first use
#ifndef FS2_PF_FENCE
#define FS2_PF_FENCE 1
#endif
// weight-fragment ring depth: 4 (three k-steps ahead; taps 1.. in a rolled loop, 8 % 4 == 0 keeps the ring index static) or 3 (two
// ahead, 16 registers fewer; the 24 k-steps of a layer fully unrolled, 24 % 3 == 0)
#ifndef FS2_PF_RING
#define FS2_PF_RING 3
#endif
constexpr int PF_RING = FS2_PF_RING;
static_assert(PF_RING == 3 || PF_RING == 4, "ring depth 3 or 4");
#if FS2_PF_FENCE
#define PF_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PF_FENCE() do { } while (0)
#endif

}  // namespace

// W: (256, taps*256) tap-major bf16 rows of one layer (taps = 3: the dense conv; 1: the pointwise half of a depth-wise layer) ->
// fragment order [tap][kb][wave][ni][lane] x 16 B.
// Wave wv owns output channels wv*32 .. +31; MFMA row i of fragment ni <-> channel
// wv*32 + (i>>2)*8 + ni*4 + (i&3), so that a lane ends up with 8 consecutive channels (one 16-byte
// slab slot) per row.
__global__ void pack_predictor_weights_kernel(const bf16* __restrict__ W, uint4* __restrict__ out, int taps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= taps * PF_KB * PF_STEP_U4) return;
    const int lane = idx & 63, ni = (idx >> 6) & 1, wv = (idx >> 7) & 7, step = idx >> 10;
    const int tap = step / PF_KB, kb = step % PF_KB, fr = lane & 15, fg = lane >> 4;
    const int ch = wv * 32 + (fr >> 2) * 8 + ni * 4 + (fr & 3);
    out[idx] = *(const uint4*)(W + (size_t)ch * (taps * PF_H) + tap * PF_H + kb * 32 + fg * 8);
}

// NWV waves, each: ALL R rows x (256 / NWV) output channels = NFR MFMA column fragments (a lane owns 8
// consecutive channels per fragment pair).  The weight fragments a wave needs are private to it (no
// redundant loads inside the workgroup) and ride a 4-deep register ring: L2 latency under 256 CUs
// pulling the same lines is ~2k cycles, a k-step is 450-900.  The activation fragments come from the
// LDS slab in two halves so that they never hold more than 16-28 VGPRs.
// The shipped form is 4 waves x 64 channels on 112-row tiles, TWO workgroups per CU (while one wave of a SIMD waits for its
// fragments the other issues), and 15 x 112 rows
// cover 1536 frames with less halo waste than 8 x 224 (5 layers: 121 us vs 137 us for the 8-wave,
// 224-row, one-workgroup-per-CU form).  One form for every shape keeps results independent of how an
// utterance is batched (the cross-wave reduction tree is part of the arithmetic).
//
// Where the time goes (r03, C2 variance predictor, 32 x 1536 frames, 5 layers; probe builds -DFS2_PF_PROBE=1|2 through
// tools/build_variant_files.sh, tools/bench_ops.py pred, kernel time = the printed figure minus ~15 us of weight packing):
//   shipped r02 form 113 us | every k-step streaming the same two weight blocks (L1-hot) -9 | no ReLU / LayerNorm / slab rewrite
//   -17 | both -35 (72 us: the K loops alone, 56 us of MFMA passes at ~1.9 GHz).  The parts ADD - the two workgroups of a CU
//   (blockIdx i and i + 256, started within 40 ticks of each other: tools/probes/hwid_probe.hip) do not hide each other's epilogue:
//   starting the second one 1-12 us late changes nothing (the delay is absorbed, no more), so the lever is instruction count.
//   Taken: the epilogue on packed fp32 (~1400 -> ~750 VALU per layer and wave), the A-fragment halves requested one half ahead,
//   the bias as the C operand of a chain's first MFMA: 113 -> 106 us.  The epilogue change alone measured nothing until the
//   fragment prefetch was in (its LDS round trips sat in front of every half's MFMAs).  hipcc still sinks most fragment reads to
//   just before their first use (a __builtin_amdgcn_sched_group_barrier pattern per k-block made that worse, not better): what is
//   left in the K loop needs the reads and their waits as asm statements, the attention_pipe.hip way - not done.
//   What did work: __builtin_amdgcn_sched_barrier(0) fences around each half's fragment reads (PF_FENCE) - reads in a group, then the
//   MFMAs, nothing moved across - once the epilogue's per-lane offsets were kept from being hoisted across the K loop (the laundered
//   lane id below; 54 -> 34 spilled registers, none inside the loops).  64-row tiles (the duration predictor's launch): 33 -> 21 us;
//   112-row tiles: 106 -> ~104 us (FS2_PF_FENCE=0 builds the unfenced form for A/B).  Then the weight ring at 3 stages instead of 4 (two
//   k-steps ahead; 16 registers fewer: 34 -> 14 spilled; the layer's 24 k-steps unrolled so that the ring index stays static):
//   ~103 -> ~98 us (-DFS2_PF_RING=4 builds the 4-stage form).
// X3 (r05; the fp32x3 / mixed3 engines' predictors): fp32 input, every product as bf16 x 3 split products - the slab holds the
// activations as bf16 heads AND tails (two planes, same swizzle; x = hi + lo to 2^-17, split_bf16x3), the weights stream as two
// fragment sets (heads, tails: PredictorArgs::wpk / wpk_lo), and a product is three MFMAs, small terms first (w_lo x_hi, w_hi x_lo,
// w_hi x_hi) - the arithmetic of the per-layer split launches (gemm_mfma.hip SPLIT) with the K sum in this kernel's order.  The
// LayerNorm epilogue splits its fp32 rows again for the next layer.  One 4-wave workgroup per CU (two planes = 117 KB), a lone wave
// per SIMD with the 512-register file: both fragment sets, both weight rings.  The embedding tail reads and writes fp32 rows.
// DW (r06; the reference's own architecture - depth-wise VarianceConvolutionLayer, model.py:541-558: Conv1d(256, 256, 3, groups = 256) ->
// Conv1d(256, 256, 1) -> ReLU -> LayerNorm): a layer is [depth-wise pass over the slab, in place] -> [ONE tap of the K loop on the
// pointwise weights] -> the same epilogue.  The depth-wise pass: a thread owns one 16-byte slot (8 channels) of R / 8 consecutive
// slab rows, walks its window once (each row unpacked once, three multiply-adds per channel in the order of rowops.hip's dwconv_kernel:
// w0 x[t-1], + w1 x[t], + w2 x[t+1], + bias, one rounding to bf16 - the bits of the per-layer launch), holds the results in registers
// across a barrier (every thread has read its window) and writes them back over the rows it read.  Rows 0 and R + 1 keep the layer-0
// input: one more stale row per layer at both ends, the dense form's halo.  A third of the dense layer's MFMA work, ~450 VALU per thread
// and layer for the depth-wise pass; ref-default's 15 + 2 layers = 4 launches instead of 34.
// Where its 67 us go (C2-sized launch, 5 layers; probe builds -DFS2_PF_PROBE=2|4|6, profiles/r06_v10_dw_predictor_ablation.txt):
// ReLU + LayerNorm epilogues 25 us, depth-wise passes 16 us, the rest (fill, five 8-step K loops = ~17 us of MFMA passes, head, tail) 28 us.
template <int MI16, int NWV, int MINW, bool X3 = false, bool DW = false>
__global__ __launch_bounds__(NWV * 64, MINW) void predictor_fused_kernel(PredictorArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
    static_assert(!(X3 && DW), "no split-arithmetic depth-wise form");
    constexpr int R = MI16 * 16, HFA = (MI16 + 1) / 2;  // row fragments: first half HFA, second MI16 - HFA
    constexpr int NFR = PF_H / (NWV * 16), NP = NFR / 2;
    constexpr int PLANE = (R + 2) * PF_ROWB;
    constexpr int TAPS = DW ? 1 : PF_TAPS, STEPS = TAPS * PF_KB;  // K loop: taps x 8 k-steps of 32
    constexpr int RING = (X3 || DW) ? 4 : PF_RING;  // X3: the 4-stage ring with the taps in a rolled loop (the unrolled 3-stage form spilled 43 registers, 35 of them inside the K loops); DW: 8 k-steps per layer, 8 % 4 == 0
    __shared__ __attribute__((aligned(16))) unsigned char slab[(X3 ? 2 : 1) * PLANE];  // slab index i <-> t = t0 - 1 + i; X3: heads, then tails
    __shared__ float red[2][NWV * R];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int S = p.S, nl = p.nlayers, halo = nl - 1, V = R - 2 * halo;
    const int tiles = (S + V - 1) / V;
    const int ub = blockIdx.x / tiles, tm = blockIdx.x % tiles;
    const int t0 = tm * V - halo;  // time of tile row 0

    if constexpr (X3) {
        // fp32 rows -> heads + tails, through registers: physical slot ps of slab row i holds logical slot ls (8 channels)
        const float* xu = (const float*)p.x + (size_t)ub * S * PF_H;
        // every load of the fill is requested before the first split (unconditional, from a clamped row; rows outside the utterance are
        // zeroed afterwards): one memory round trip instead of fifteen
        constexpr int NIT = ((R + 2) * 32 + NWV * 64 - 1) / (NWV * 64);
        uint4 c0[NIT], c1[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            int sidx = tid + u * NWV * 64;
            sidx = sidx < (R + 2) * 32 ? sidx : (R + 2) * 32 - 1;
            const int i = sidx >> 5, ps = sidx & 31, t = t0 - 1 + i;
            const int ls = (ps & 16) | ((((ps & 7) ^ (i & 7)) << 1) | ((ps >> 3) & 1));
            const int tc = t < 0 ? 0 : (t < S ? t : S - 1);
            c0[u] = *(const uint4*)(xu + (size_t)tc * PF_H + ls * 8);
            c1[u] = *(const uint4*)(xu + (size_t)tc * PF_H + ls * 8 + 4);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int sidx = tid + u * NWV * 64;
            if (sidx < (R + 2) * 32) {
                const int i = sidx >> 5, ps = sidx & 31, t = t0 - 1 + i;
                uint4 hi, lo;
                split_bf16x3(c0[u], c1[u], hi, lo);
                if (t < 0 || t >= S) hi = lo = make_uint4(0u, 0u, 0u, 0u);
                *(uint4*)(slab + i * PF_ROWB + (ps << 4)) = hi;
                *(uint4*)(slab + PLANE + i * PF_ROWB + (ps << 4)) = lo;
            }
        }
    } else
    // ---- slab fill: 2 rows (1 KiB) per DMA instruction, 16-byte XOR swizzle applied on the source side
    {
        const bf16* xu = (const bf16*)p.x + (size_t)ub * S * PF_H;
        const __amdgpu_buffer_rsrc_t xrs =
            __builtin_amdgcn_make_buffer_rsrc((void*)xu, 0, (unsigned)((size_t)S * PF_ROWB), 0x00020000);
        constexpr int NCH = (R + 2) / 2;
#pragma unroll
        for (int k = 0; k < (NCH + NWV - 1) / NWV; ++k) {
            const int c = k * NWV + wv;
            if (c < NCH) {
                const int i = 2 * c + (lane >> 5), ps = lane & 31, t = t0 - 1 + i;
                // physical slot ps of slab row i holds logical slot ls (the inverse of SlabSwizzle::slot, fs2_common.h)
                const int ls = (ps & 16) | ((((ps & 7) ^ (i & 7)) << 1) | ((ps >> 3) & 1));
                const unsigned voff = (t >= 0 && t < S) ? (unsigned)(t * PF_ROWB + (ls << 4)) : 0xFFFFF000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(slab + c * 1024),
                                                         16, voff, 0, 0, 0);
            }
        }
    }

    // ---- weight stream: step g = (layer*3 + tap)*8 + kb, 16 KiB per step, this wave's NFR fragments
    // (fragment order [step][32-channel group][2][lane]: a wave's fragments are contiguous for any NWV)
    const uint4* __restrict__ wbase = (const uint4*)p.wpk + wv * NFR * 64 + lane;
    const uint4* __restrict__ wbase_lo = X3 ? (const uint4*)p.wpk_lo + wv * NFR * 64 + lane : nullptr;
    const int total = nl * STEPS;
    auto loadB = [&](uint4 (&b)[NFR], int g) {
        g = g < total ? g : total - 1;  // past the end: a harmless re-read instead of a branch
#if defined(FS2_PF_PROBE) && (FS2_PF_PROBE & 1)
        g &= 1;  // probe: every k-step streams the same two 16-KiB blocks (hot in the vector L1) - what does the L2 weight stream cost?
#endif
#pragma unroll
        for (int ni = 0; ni < NFR; ++ni) b[ni] = wbase[(size_t)g * PF_STEP_U4 + ni * 64];
    };
    auto loadBL = [&](uint4 (&b)[NFR], int g) {  // X3: the tails' stream
        g = g < total ? g : total - 1;
#pragma unroll
        for (int ni = 0; ni < NFR; ++ni) b[ni] = wbase_lo[(size_t)g * PF_STEP_U4 + ni * 64];
    };
    uint4 bw[RING][NFR], bwl[X3 ? RING : 1][NFR];
    loadB(bw[0], 0);
    loadB(bw[1], 1);
    if constexpr (RING == 4) loadB(bw[2], 2);
    if constexpr (X3) {
        loadBL(bwl[0], 0);
        loadBL(bwl[1], 1);
        if constexpr (RING == 4) loadBL(bwl[2], 2);
    }

    dma_drain();
    __syncthreads();

    const int n0 = wv * (NFR * 16) + fg * 8;  // fragment pair j: this lane's channels n0 + 32*j .. +7
    for (int l = 0; l < nl; ++l) {
#if defined(FS2_PF_PROBE) && (FS2_PF_PROBE & 4)
        if constexpr (false) {  // probe: no depth-wise pass - what does it cost?
#else
        if constexpr (DW) {
#endif
            // ---- depth-wise Conv1d(k = 3, groups = 256) over the slab, in place (model.py:545-551) ----
            static_assert(NWV * 64 == 256 && R % 8 == 0, "depth-wise pass: 32 slots x 8 row groups");
            constexpr int RPT = R / 8;  // rows per thread
            int tid_d = tid;
            asm volatile("" : "+v"(tid_d));  // (addresses re-derived per layer, not carried across the K loop)
            const int L = tid_d & 31, i0 = 1 + (tid_d >> 5) * RPT;  // logical slot (channels 8 L ..), first slab row
            f32x2_t wt[3][4], bs[4];
            {
                const float* wp = p.dw_w + (size_t)l * 3 * PF_H + L * 8;
#pragma unroll
                for (int tp = 0; tp < 3; ++tp) {
                    const float4 a = *(const float4*)(wp + tp * PF_H), b = *(const float4*)(wp + tp * PF_H + 4);
                    wt[tp][0] = (f32x2_t){a.x, a.y}; wt[tp][1] = (f32x2_t){a.z, a.w}; wt[tp][2] = (f32x2_t){b.x, b.y}; wt[tp][3] = (f32x2_t){b.z, b.w};
                }
                const float* bp = p.dw_b + (size_t)l * PF_H + L * 8;
                const float4 a = *(const float4*)bp, b = *(const float4*)(bp + 4);
                bs[0] = (f32x2_t){a.x, a.y}; bs[1] = (f32x2_t){a.z, a.w}; bs[2] = (f32x2_t){b.x, b.y}; bs[3] = (f32x2_t){b.z, b.w};
            }
            const SlabSwizzle sw(PF_ROWB / 16);
            auto rowp = [&](int i) { return slab + i * PF_ROWB + (sw.slot(L, i) << 4); };
            auto unpack = [](const uint4& v, f32x2_t (&f)[4]) {
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = (f32x2_t){__uint_as_float(u[q] << 16), __uint_as_float(u[q] & 0xffff0000u)};
            };
            uint4 raw[RPT + 2];
#pragma unroll
            for (int j = 0; j < RPT + 2; ++j) raw[j] = *(const uint4*)rowp(i0 - 1 + j);
            uint4 o[RPT];
            f32x2_t xa[4], xb[4], xc[4];
            unpack(raw[0], xa);
            unpack(raw[1], xb);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                unpack(raw[j + 2], xc);
                unsigned w4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x2_t a = wt[0][q] * xa[q];  // = fma(w0, x, 0): dwconv_kernel's chain starts from a zero accumulator
                    a = __builtin_elementwise_fma(wt[1][q], xb[q], a);
                    a = __builtin_elementwise_fma(wt[2][q], xc[q], a);
                    a = a + bs[q];
                    w4[q] = pack_bf16x2(a.x, a.y);
                    xa[q] = xb[q];
                    xb[q] = xc[q];
                }
                o[j] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            __syncthreads();  // every window has been read
#pragma unroll
            for (int j = 0; j < RPT; ++j) *(uint4*)rowp(i0 + j) = o[j];
            __syncthreads();
        }
        // The bias rides in as the C operand of a chain's FIRST MFMA (k-block 0 of tap 0 names the bias quad, the accumulator is
        // written, not read: no 112 v_mov per layer to seed the accumulators) - tap 0 is peeled for that.
        f32x4_t acc[NFR][MI16];
        f32x4_t bq[NFR];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const float* bias = p.bias + l * PF_H + n0 + 32 * j;
            const float4 b0 = *(const float4*)bias, b1 = *(const float4*)(bias + 4);
            bq[2 * j] = (f32x4_t){b0.x, b0.y, b0.z, b0.w};
            bq[2 * j + 1] = (f32x4_t){b1.x, b1.y, b1.z, b1.w};
        }
        // tile row r at tap tp lives at slab index r + tp; its 16-byte slot (kb*4 + fg) is stored at
        // SlabSwizzle::slot(slot, index) (conflict-free fragment reads for every start row; the plain
        // slot ^ (index & 15) map measured 20 % bank-conflict cycles).  Taps 1.. : a rolled loop (short live ranges); the 8 k-blocks of
        // a tap are unrolled so that the ring index is static.
        auto mma = [&](auto FIRSTKB, const uint4& bfrag, const uint4& afrag, int ni, f32x4_t& a) {
            if constexpr (decltype(FIRSTKB)::value)
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&bfrag, *(const bf16x8_t*)&afrag, bq[ni], 0, 0, 0);
            else
                Mma16<bf16>::step(bfrag, afrag, a);
        };
        // X3: (w_lo x_hi) + (w_hi x_lo) + (w_hi x_hi), small terms first; the bias rides in the first of the three
        auto mma3 = [&](auto FIRSTKB, const uint4& bh, const uint4& bl, const uint4& ah, const uint4& al, int ni, f32x4_t& a) {
            mma(FIRSTKB, bl, ah, ni, a);
            Mma16<bf16>::step(bh, al, a);
            Mma16<bf16>::step(bh, ah, a);
        };
        auto tap = [&](int tp, auto FIRST) {
            constexpr bool first = decltype(FIRST)::value;
            const int i0 = fr + tp + (DW ? 1 : 0);  // DW: the pointwise conv reads its own row
            const unsigned char* arow_p = slab + i0 * PF_ROWB;
            const int acx = (((fg & 1) << 3) | ((fg >> 1) ^ (i0 & 7))) << 4;  // SlabSwizzle::slot(fg, i0); + kb below
            constexpr int HFB = MI16 - HFA;
            auto loadA = [&](uint4 (&fx)[HFA], int kb, int hf) {
                const int m0 = hf * HFA, cnt = hf ? HFB : HFA;
#pragma unroll
                for (int mi = 0; mi < HFA; ++mi)
                    if (mi < cnt) fx[mi] = *(const uint4*)(arow_p + (m0 + mi) * 16 * PF_ROWB + (acx ^ ((((kb & 3) << 1) | ((kb >> 2) << 4)) << 4)));
            };
            auto loadAL = [&](uint4 (&fx)[HFA], int kb, int hf) {  // X3: the tails' plane
                const int m0 = hf * HFA, cnt = hf ? HFB : HFA;
#pragma unroll
                for (int mi = 0; mi < HFA; ++mi)
                    if (mi < cnt) fx[mi] = *(const uint4*)(arow_p + PLANE + (m0 + mi) * 16 * PF_ROWB + (acx ^ ((((kb & 3) << 1) | ((kb >> 2) << 4)) << 4)));
            };
            auto kblock = [&](auto KB0, int kb, uint4 (&fxa)[HFA], uint4 (&fxb)[HFA], uint4 (&fla)[HFA], uint4 (&flb)[HFA]) {
                const int rs = (tp * PF_KB + kb) % RING, rn = (tp * PF_KB + kb + RING - 1) % RING;  // static after unrolling
                loadB(bw[rn], (l * TAPS + tp) * PF_KB + kb + RING - 1);
                if constexpr (X3) {
                    // row block by row block: one (head, tail) fragment pair and its 3 NFR MFMAs (192 cycles), the next pair requested
                    // one block ahead (a two-deep ring in fxa[0..1] / fla[0..1], indexed by the running block count - static after
                    // unrolling): 16 fragment registers instead of the 64 of two half-tile sets of both planes
                    loadBL(bwl[rn], (l * TAPS + tp) * PF_KB + kb + RING - 1);
#pragma unroll
                    for (int mi = 0; mi < MI16; ++mi) {
                        const int c = kb * MI16 + mi, cur = c & 1, nxt = cur ^ 1;
                        const int nmi = mi + 1 < MI16 ? mi + 1 : 0, nkb = mi + 1 < MI16 ? kb : kb + 1;
                        if (nkb < PF_KB) {
                            const int off = nmi * 16 * PF_ROWB + (acx ^ ((((nkb & 3) << 1) | ((nkb >> 2) << 4)) << 4));
                            fxa[nxt] = *(const uint4*)(arow_p + off);
                            fla[nxt] = *(const uint4*)(arow_p + PLANE + off);
                        }
                        PF_FENCE();
#pragma unroll
                        for (int ni = 0; ni < NFR; ++ni) mma3(KB0, bw[rs][ni], bwl[rs][ni], fxa[cur], fla[cur], ni, acc[ni][mi]);
                        PF_FENCE();
                    }
                    (void)fxb; (void)flb;
                } else if constexpr (MINW == 1 || PF_PREFETCH) {
                    // the next half's activation fragments are requested before this half's MFMAs (two fragment sets alive): an LDS
                    // round trip behind every half otherwise (r03: 113 -> 107 us for the C2 variance predictor with two workgroups per CU)
                    loadA(fxb, kb, 1);
                    PF_FENCE();
#pragma unroll
                    for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
                        for (int mi = 0; mi < HFA; ++mi) mma(KB0, bw[rs][ni], fxa[mi], ni, acc[ni][mi]);
                    PF_FENCE();
                    if (kb + 1 < PF_KB) loadA(fxa, kb + 1, 0);
                    PF_FENCE();
#pragma unroll
                    for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
                        for (int mi = 0; mi < HFB; ++mi) mma(KB0, bw[rs][ni], fxb[mi], ni, acc[ni][HFA + mi]);
                    PF_FENCE();
                } else {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int m0 = hf * HFA, cnt = hf ? HFB : HFA;
                        loadA(fxa, kb, hf);
#pragma unroll
                        for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
                            for (int mi = 0; mi < HFA; ++mi)
                                if (mi < cnt) mma(KB0, bw[rs][ni], fxa[mi], ni, acc[ni][m0 + mi]);
                    }
                }
            };
            uint4 fxa[HFA], fxb[HFA], fla[X3 ? HFA : 1], flb[X3 ? HFA : 1];
            if constexpr (X3) {  // the tap's first fragment pair (ring slot 0)
                fxa[0] = *(const uint4*)(arow_p + acx);
                fla[0] = *(const uint4*)(arow_p + PLANE + acx);
            } else if constexpr (MINW == 1 || PF_PREFETCH) {
                loadA(fxa, 0, 0);
            }
            if constexpr (X3) {
                kblock(std::integral_constant<bool, first>{}, 0, fxa, fxb, fla, flb);
#pragma unroll
                for (int kb = 1; kb < PF_KB; ++kb) kblock(std::false_type{}, kb, fxa, fxb, fla, flb);
            } else {
                uint4 (&d0)[HFA] = fxa, (&d1)[HFA] = fxb;  // (unused tails' sets)
                kblock(std::integral_constant<bool, first>{}, 0, fxa, fxb, d0, d1);
#pragma unroll
                for (int kb = 1; kb < PF_KB; ++kb) kblock(std::false_type{}, kb, fxa, fxb, d0, d1);
            }
        };
        tap(0, std::true_type{});
        if constexpr (RING == 4) {
#pragma unroll 1
            for (int tp = 1; tp < TAPS; ++tp) tap(tp, std::false_type{});
        } else {
            tap(1, std::false_type{});
            tap(2, std::false_type{});
        }

        // ---- ReLU + LayerNorm (r03: packed fp32) ----  lane: rows (m*16 + fr), channels n0 + 32*(ni>>1) + (ni&1)*4 + r.
        // The epilogue's VALU stream runs NEXT to the co-resident workgroup's MFMAs on the same SIMD and the two add up (profiles/HISTORY.md §4):
        // before, ~1400 VALU per layer and wave against 672 MFMAs (fmaxf on an MFMA result = a canonicalising v_max + the v_max,
        // scalar adds / subtracts / squares, an IEEE 1/sqrt of ~35 instructions per row).  Now: one v_max per element, the row
        // sums / centring / squares on v_pk_add_f32 / v_pk_fma_f32 (two elements per instruction; the partial sums pair up as
        // (even, odd) slots of a fragment), v_rsq_f32 for the reciprocal root (1 ulp; the per-layer path keeps the IEEE form).
#if defined(FS2_PF_PROBE) && (FS2_PF_PROBE & 2)
        {   // probe: no ReLU / LayerNorm / slab rewrite - the K loops alone (the sum keeps the MFMAs alive)
            f32x4_t s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
                for (int m = 0; m < MI16; ++m) s4 += acc[ni][m];
            if (s4.x + s4.y + s4.z + s4.w == 12345.678f) red[0][tid] = s4.x;
            __syncthreads();
            continue;
        }
#endif
        auto relu = [](float x) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; };
        // The epilogue's per-lane offsets (slab slots, LayerNorm parameter columns, exchange rows) are re-derived from a laundered
        // lane id: left to itself hipcc hoists them out of the layer loop and carries ~40 registers of addresses across the K loop,
        // which at 256 registers is what spills once the fragment reads are pinned one half ahead (54 spilled, +4 us).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int fr = lane_e & 15, fg = lane_e >> 4;
        const int n0 = wv * (NFR * 16) + fg * 8;
        const float invn = 1.0f / (float)PF_H;
        float mean[MI16], rstd[MI16];
#pragma unroll
        for (int m = 0; m < MI16; ++m) {
            f32x2_t s2 = {0.f, 0.f};
#pragma unroll
            for (int ni = 0; ni < NFR; ++ni) {
                f32x4_t& a = acc[ni][m];
                a = (f32x4_t){relu(a[0]), relu(a[1]), relu(a[2]), relu(a[3])};
                s2 += a.xy;
                s2 += a.zw;
            }
            const float sm = group4_sum(s2.x + s2.y);
            if (fg == 0) red[0][wv * R + m * 16 + fr] = sm;
        }
        __syncthreads();  // also: every wave is past its K loop -> the slab may be rewritten below
#pragma unroll
        for (int m = 0; m < MI16; ++m) {
            const int row = m * 16 + fr;
            float t8 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t8 += red[0][w * R + row];
            mean[m] = t8 * invn;
            const f32x2_t mn2 = {mean[m], mean[m]};
            f32x2_t q2 = {0.f, 0.f};
#pragma unroll
            for (int ni = 0; ni < NFR; ++ni) {
                f32x4_t& a = acc[ni][m];
                const f32x2_t d0 = a.xy - mn2, d1 = a.zw - mn2;  // kept: the normalisation below needs the same difference
                a = (f32x4_t){d0.x, d0.y, d1.x, d1.y};
                q2 = __builtin_elementwise_fma(d0, d0, q2);
                q2 = __builtin_elementwise_fma(d1, d1, q2);
            }
            const float q = group4_sum(q2.x + q2.y);
            if (fg == 0) red[1][wv * R + row] = q;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MI16; ++m) {
            const int row = m * 16 + fr;
            float t8 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t8 += red[1][w * R + row];
            rstd[m] = __builtin_amdgcn_rsqf(t8 * invn + p.eps);
        }
        const bool last = l + 1 == nl;
        float dsum[MI16];
#pragma unroll
        for (int m = 0; m < MI16; ++m) dsum[m] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int n = n0 + 32 * j;
            f32x2_t gg[4], ee[4], hw[4];
            {
                const float* gam = p.ln_g + l * PF_H + n;
                const float* bet = p.ln_b + l * PF_H + n;
                const float4 g0 = *(const float4*)gam, g1 = *(const float4*)(gam + 4);
                const float4 e0 = *(const float4*)bet, e1 = *(const float4*)(bet + 4);
                gg[0] = (f32x2_t){g0.x, g0.y}; gg[1] = (f32x2_t){g0.z, g0.w}; gg[2] = (f32x2_t){g1.x, g1.y}; gg[3] = (f32x2_t){g1.z, g1.w};
                ee[0] = (f32x2_t){e0.x, e0.y}; ee[1] = (f32x2_t){e0.z, e0.w}; ee[2] = (f32x2_t){e1.x, e1.y}; ee[3] = (f32x2_t){e1.z, e1.w};
                if (last) {
                    const float4 h0 = *(const float4*)(p.head_w + n), h1 = *(const float4*)(p.head_w + n + 4);
                    hw[0] = (f32x2_t){h0.x, h0.y}; hw[1] = (f32x2_t){h0.z, h0.w}; hw[2] = (f32x2_t){h1.x, h1.y}; hw[3] = (f32x2_t){h1.z, h1.w};
                }
            }
#pragma unroll
            for (int m = 0; m < MI16; ++m) {
                const int row = m * 16 + fr, t = t0 + row, i = row + 1;
                const f32x2_t rs2 = {rstd[m], rstd[m]};
                f32x2_t y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4_t& a = acc[2 * j + (r >> 1)][m];
                    y[r] = __builtin_elementwise_fma((r & 1 ? a.zw : a.xy) * rs2, gg[r], ee[r]);
                }
                if (last) {
                    f32x2_t d2 = {dsum[m], 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2 = __builtin_elementwise_fma(y[r], hw[r], d2);
                    dsum[m] = d2.x + d2.y;
                } else {
                    // next layer's input, in place; rows outside the utterance stay the conv's zero padding
                    const bool inside = t >= 0 && t < S;
                    const uint4 o = inside ? make_uint4(pack_bf16x2(y[0].x, y[0].y), pack_bf16x2(y[1].x, y[1].y),
                                                        pack_bf16x2(y[2].x, y[2].y), pack_bf16x2(y[3].x, y[3].y))
                                           : make_uint4(0u, 0u, 0u, 0u);
                    const int so = i * PF_ROWB + (SlabSwizzle(PF_ROWB / 16).slot(n >> 3, i) << 4);
                    *(uint4*)(slab + so) = o;
                    if constexpr (X3) {  // the tails: RNE(y - head), the subtraction exact in fp32 (split_bf16x3)
                        const unsigned hw4[4] = {o.x, o.y, o.z, o.w};
                        unsigned lw[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            lw[r] = pack_bf16x2(y[r].x - __uint_as_float(hw4[r] << 16), y[r].y - __uint_as_float(hw4[r] & 0xffff0000u));
                        *(uint4*)(slab + PLANE + so) = inside ? make_uint4(lw[0], lw[1], lw[2], lw[3]) : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
        }
        if (!last) {
            __syncthreads();  // slab holds layer l's output
            continue;
        }
        // ---- Linear(256, 1) head + mask (model.py:519-522) ----
#pragma unroll
        for (int m = 0; m < MI16; ++m) {
            const float d = group4_sum(dsum[m]);
            if (fg == 0) red[0][wv * R + m * 16 + fr] = d;  // red[0] was last read two barriers ago
        }
        __syncthreads();
        if (tid < R) {
            const int row = tid, t = t0 + row;
            if (row >= halo && row < R - halo && t < S) {
                float d = p.head_b;
#pragma unroll
                for (int w = 0; w < NWV; ++w) d += red[0][w * R + row];
                const size_t o = (size_t)ub * S + t;
                d = (p.mask && p.mask[o]) ? 0.f : d;
                p.pred[o] = d;
                red[1][row] = d;  // (red[1] was last read before the final layer's last barrier)
            }
        }
    }
    // (outside the layer loop: inside it hipcc carried the tail's addresses and edge registers across every K loop - 116 spilled)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    if (p.be_y) {
        // ---- VarianceEncoder tail: y[t] = x[t] + Emb[bucketize(pred[t] * std + mean)] [+ pe[t]] [+ spk[b]] for the finished rows.
        // bucketize = the number of edges < v: the wave holds the edges in registers (8 per lane) and counts by ballot; a lane
        // owns 4 consecutive channels of a row; EB rows' loads are in flight at once (x comes back out of L2 / MALL).
        __syncthreads();
        constexpr int EB = 8;  // rows in flight per wave (13 - two even rounds of the 104 finished rows - measured no better)
        const int nedge = p.be_nbins - 1, nb = (nedge + 63) >> 6;
        float bl[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = lane_t + 64 * k;
            bl[k] = i < nedge ? p.be_bins[i] : __builtin_inff();
        }
        using XT = typename std::conditional<X3, float, bf16>::type;  // X3: fp32 rows in and out
        using XV = typename std::conditional<X3, float4, uint2>::type;
        const XT* xg = (const XT*)p.x + (size_t)ub * S * PF_H + lane_t * 4;
        XT* yg = (XT*)p.be_y + (size_t)ub * S * PF_H + lane_t * 4;
        const float* spr = p.be_spk ? p.be_spk + (size_t)ub * PF_H + lane_t * 4 : nullptr;
        float4 sp4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (spr) sp4 = *(const float4*)spr;
        for (int base = halo + wv * EB; base < R - halo; base += NWV * EB) {
            XV xv[EB];
            float4 ev[EB], pv[EB];
            bool okr[EB];
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int row = base + j, t = t0 + row;
                okr[j] = row < R - halo && t < S;
                const int rc = okr[j] ? row : halo, tc = okr[j] ? t : t0 + halo;
                const float v = __fadd_rn(__fmul_rn(red[1][rc], p.be_std), p.be_mean);
                int lo = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < nb) lo += __popcll(__ballot(bl[k] < v));
                xv[j] = *(const XV*)(xg + (size_t)tc * PF_H);
                ev[j] = *(const float4*)(p.be_emb + (size_t)lo * PF_H + lane_t * 4);
                if (p.be_pe) pv[j] = *(const float4*)(p.be_pe + (size_t)tc * PF_H + lane_t * 4);
            }
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                if (!okr[j]) continue;
                float v[4];
                if constexpr (X3) {
                    v[0] = xv[j].x; v[1] = xv[j].y; v[2] = xv[j].z; v[3] = xv[j].w;
                } else {
                    v[0] = __uint_as_float(xv[j].x << 16); v[1] = __uint_as_float(xv[j].x & 0xffff0000u);
                    v[2] = __uint_as_float(xv[j].y << 16); v[3] = __uint_as_float(xv[j].y & 0xffff0000u);
                }
                const float e4[4] = {ev[j].x, ev[j].y, ev[j].z, ev[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], e4[i]);
                if (p.be_pe) {
                    const float q4[4] = {pv[j].x, pv[j].y, pv[j].z, pv[j].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], q4[i]);
                }
                if (spr) {
                    const float q4[4] = {sp4.x, sp4.y, sp4.z, sp4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], q4[i]);
                }
                if constexpr (X3) *(float4*)(yg + (size_t)(t0 + base + j) * PF_H) = make_float4(v[0], v[1], v[2], v[3]);
                else *(uint2*)(yg + (size_t)(t0 + base + j) * PF_H) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
#else
    (void)p;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two tiles per workgroup, one wave per SIMD (r03): the LayerNorm epilogue of one tile is issued BETWEEN the MFMAs of the other
// tile's K loop.  In the kernel above a layer is [672 MFMAs] [~1300 VALU of ReLU / LayerNorm / pack] per wave, and the second
// workgroup on the CU does not fill the VALU phase with its MFMAs (the issue arbiter serves the older wave; s_setprio changes
// nothing: profiles/HISTORY.md §4) - MFMA pipe 0.48 busy.  Here a 4-wave workgroup owns tiles A and B (two 114-row slabs in LDS, both
// accumulator sets in the 512-register file) and runs
//     K(A,0) | K(B,0) + E(A,0) | K(A,1) + E(B,0) | K(B,1) + E(A,1) | ... | K(B,n-1) + E(A,n-1) | E(B,n-1)
// where K + E means: one MFMA, then the next one or two instructions of the other tile's epilogue, then the next MFMA (two
// plain VALU hide under a 16x16x32 MFMA, tools/probes/mfma16_filler_cost.hip).  The epilogue is cut into three stages, one per
// conv tap of the K loop it rides in (row sums | centred squares | normalise + pack + slab store), each ending in the barrier
// its LDS exchange needs.  Per-row arithmetic and its order are those of predictor_fused_kernel: bit-identical outputs.
// MEASURED (C2 variance predictor, 32 x 1536 frames, 5 layers): 120.4 us against 112-114 us for the kernel above - not shipped,
// knob 1302 selects it.  Segment stamps (tools/probes/pred_pair_stamps.py, ticks of s_memtime; 672 MFMAs = 12.9 k ticks):
// K alone (cold) 24.7 k, K + E 20.3-22.3 k each, last K + E (head dots) 31.9 k, E alone (the drain) 43.5 k; inside a K + E
// segment the three stages cost 6.0 k / 6.4 k / 8.2 k against 4.3 k of MFMAs.  So the epilogue does hide (K + E costs no more than
// K), but the K loop itself runs at 22-30 cycles per 16-cycle MFMA: its own loads, address arithmetic and waits (~1 instruction
// per gap) plus two epilogue instructions per gap are more than a 16x16x32 gap absorbs, and fill + drain are a quarter of the
// kernel.  The form that can pass 0.45 of peak is this schedule on 32x32x16 MFMAs (32-cycle gaps, five fillers each, as the
// attention kernel) - in EVERY predictor variant at once, because the MFMA shape enters the rounding (profiles/HISTORY.md §4 "Round 3").
#ifdef FS2_PRED_PROBE  // tools/probes/pred_pair_stamps.py: s_memtime of workgroup 0 / wave 0 at every segment boundary
__device__ unsigned long long g_pred_stamps[64];
#define PRED_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pred_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PRED_STAMP(i) do { } while (0)
#endif
// (r03: two further forms of this kernel were built, bit-identical, measured slower and - r05 - removed: 208-row tiles on one
// 4-wave workgroup per CU (148-152 us against 112 for two 112-row workgroups per CU on the C2 variance predictor: with one wave per
// SIMD nothing runs under the LayerNorm epilogue, whose VALU stream is as long as the layer's MFMA stream) and a one-wave-per-SIMD
// workgroup that owns two 112-row tiles and issues one tile's epilogue between the other's MFMAs (120.4 us: the K loop itself runs
// at 22-30 cycles per 16-cycle MFMA and a 16x16x32 gap absorbs about two fillers).  profiles/HISTORY.md §4 "Round 3" keeps the measurements.)
bool predictor_fused_supported(int dtype, int H, int taps, int nlayers, int S) {
    return dtype == FS2_BF16 && H == PF_H && taps == PF_TAPS && nlayers >= 1 && nlayers <= 16 && S >= 1 &&
           (size_t)S * PF_ROWB < 0xFFFFF000ull;
}

int launch_pack_predictor_weights(const void* w_layer, void* out_layer, hipStream_t stream, int taps) {
    if (taps != 1 && taps != PF_TAPS) return FS2_ERR_SHAPE;
    const int n = taps * PF_KB * PF_STEP_U4;
    hipLaunchKernelGGL(pack_predictor_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const bf16*)w_layer,
                       (uint4*)out_layer, taps);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

size_t predictor_packed_bytes_per_layer(int taps) { return (size_t)taps * PF_KB * PF_STEP_U4 * 16; }

// the split-arithmetic form (fp32 rows in, PredictorArgs::wpk = the weights' bf16 heads, wpk_lo their tails; no embedding tail)
bool predictor_fused_x3_supported(int H, int taps, int nlayers, int S) {
    return H == PF_H && taps == PF_TAPS && nlayers >= 1 && nlayers <= 16 && S >= 1 && 112 - 2 * (nlayers - 1) >= 32;
}
int launch_predictor_fused_x3(const PredictorArgs& a, hipStream_t stream) {
    if (!predictor_fused_x3_supported(a.H, a.taps, a.nlayers, a.S) || !a.wpk_lo) return FS2_ERR_SHAPE;
    if (a.be_y && (a.be_y == a.x || !a.be_bins || !a.be_emb || a.be_nbins < 2 || a.be_nbins - 1 > 512)) return FS2_ERR_ARG;
    if (a.B <= 0) return FS2_OK;
    const int halo2 = 2 * (a.nlayers - 1);
    auto tiles = [&](int R) { return (long)a.B * ((a.S + (R - halo2) - 1) / (R - halo2)); };
    // (one tile height for every shape: the wave layout, not the height, enters the arithmetic - but a shorter tile is another
    //  instantiation to carry; 112 rows x 2 planes = 117 KB, one workgroup per CU)
    hipLaunchKernelGGL((predictor_fused_kernel<7, 4, 1, true>), dim3((unsigned)tiles(112)), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

int launch_predictor_fused(const PredictorArgs& a, hipStream_t stream) {
    if (!predictor_fused_supported(FS2_BF16, a.H, a.taps, a.nlayers, a.S)) return FS2_ERR_SHAPE;
    if (a.B <= 0) return FS2_OK;
    const int halo2 = 2 * (a.nlayers - 1);
    if (112 - halo2 < 32) return FS2_ERR_SHAPE;
    auto tiles = [&](int R) { return (long)a.B * ((a.S + (R - halo2) - 1) / (R - halo2)); };
    // the tile HEIGHT does not enter the arithmetic (rows are independent), the wave layout does: one
    // layout for everything, shorter tiles when 112-row tiles would leave most CUs without work
    if (a.be_y && (a.be_y == a.x || !a.be_bins || !a.be_emb || a.be_nbins < 2 || a.be_nbins - 1 > 512)) return FS2_ERR_ARG;
    if ((a.dw_w != nullptr) != (a.dw_b != nullptr)) return FS2_ERR_ARG;
    // the kernel reads the per-layer vectors 16 bytes at a time (float4 / uint4): every pointer 16-byte aligned
    if (((uintptr_t)a.x | (uintptr_t)a.wpk | (uintptr_t)a.bias | (uintptr_t)a.ln_g | (uintptr_t)a.ln_b | (uintptr_t)a.head_w | (uintptr_t)a.dw_w |
         (uintptr_t)a.dw_b | (uintptr_t)a.be_y | (uintptr_t)a.be_emb | (uintptr_t)a.be_pe | (uintptr_t)a.be_spk) & 15)
        return FS2_ERR_ARG;
    const bool small = tiles(112) < 200 && 64 - halo2 >= 32;
    if (a.dw_w) {  // depth-wise layers: wpk holds the pointwise weights (one tap)
        if (small) hipLaunchKernelGGL((predictor_fused_kernel<4, 4, 2, false, true>), dim3((unsigned)tiles(64)), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((predictor_fused_kernel<7, 4, 2, false, true>), dim3((unsigned)tiles(112)), dim3(256), 0, stream, a);
    } else if (small) {
        hipLaunchKernelGGL((predictor_fused_kernel<4, 4, 2>), dim3((unsigned)tiles(64)), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((predictor_fused_kernel<7, 4, 2>), dim3((unsigned)tiles(112)), dim3(256), 0, stream, a);
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2

