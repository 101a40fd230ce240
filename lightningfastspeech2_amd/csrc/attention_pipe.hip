// Software-pipelined fused attention for gfx950, bf16, head dim 128: the MFMA-bound instance of the attention inside
// nn.MultiheadAttention (ConformerEncoderLayer.forward, /root/reference/litfass/fastspeech2/model.py:108-116) - the decoder's
// self-attention over T frames.  Same contract as attention.hip (padded KEYS masked, padded queries computed, base-2 softmax,
// fp32 statistics, bf16 P), different schedule and no running max:
//
//   attention.hip runs a KV tile as  [16 Q.K^T MFMAs] [softmax VALU] [16 P.V MFMAs]  per wave: the matrix pipe idles while the
//   ~200 softmax instructions issue and the issue port idles while MFMAs drain (PMC: MFMA pipe 44 % busy, 5 VALU per MFMA), and
//   a second wave on the SIMD does not fill the holes (the arbiter serves the oldest wave; profiles/HISTORY.md §4).  Here one wave per SIMD
//   runs [1 MFMA, <= 5 other instructions] over and over (an in-order wave gets issue slots only in the shadow of its own last
//   MFMA; tools/probes/mfma_filler_cost.hip), and every MFMA phase carries the softmax "units" of ANOTHER 32-key half tile:
//
//     phase QK(h+1):  S(h+1) = K(h+1) Q^T  [8 K fragments x NQB MFMAs]  ||  second half of the units of S(h): p = exp2(S), row
//                                                                            sums, bf16 pack  ||  V(h) fragments (ds_read_b64_tr_b16)
//     phase PV(h):    O^T += V(h)^T P(h)^T [8 V fragments x NQB MFMAs]  ||  first half of the units of S(h+1)
//                                                                        ||  K(h+2) fragments (ds_read_b128)  ||  tile DMA pieces
//
//   A wave owns NQB 32-query blocks that share every K / V fragment it reads; two S half tiles and two P half tiles are alive
//   at a time.  p = exp2(scaled score) with no reference in the first pass (bf16 keeps 8 bits at any scale, O and the
//   denominator accumulate in fp32); a row whose denominator leaves [2^-100, 2^100] is not stored and the item runs again for it
//   from the reference log2(denominator), which then rides in the C operand of the first MFMA of each chain - so the loop has no
//   max, no rescale, no branch, and no VALU access to O (run_item's pass loop; per-row, launch-shape independent).
//   K / V tiles (64 keys) stream by buffer-load-to-LDS DMA into three K slots and two V slots: ONE barrier per tile - at the top of
//   tile t everything tile t reads (V(t), K(t+1)) has landed, K(t-1) / V(t-1) are dead and K(t+2) / V(t+1) are issued into their
//   slots, a whole tile ahead of their first read; fragments pass through 4-deep register rings.
//
// Work split: a unit is 128 queries of one (utterance, head); the grid is one workgroup per CU (NQB = 2 / 3, 512 registers, one
// wave per SIMD) or two (NQB = 1), each owning a contiguous run of units - triples of units of one head run as 384-query items
// (96 per wave, kernel<3>), pairs as 256-query items (64 per wave), leftovers as 128-query items - so 64 x 1536 queries (C2
// decoder) are 3 units = one 384-query item per workgroup, one round, no tail.
// Workgroups of one XCD own neighbouring units: a head's K / V stay in that XCD's L2.
#include "fs2_common.h"
#include "fs2_kernels.h"
#include <type_traits>

namespace fs2 {
namespace {

// Probe builds (tools/probes/attn_pipe_probe.py compiles this file with -DFS2_ATTN_PROBE=bits into throw-away libraries and times
// them): 1 = no exp / sum / pack, 2 = no row max / decision, 4 = no fragment reads in the loop, 8 = no tile DMA in the loop,
// 16 = no Q.K^T MFMAs, 32 = no P.V MFMAs, 128 = 32-query items only, 256 = no per-tile drain + barrier (racy), 512 = no key-mask
// test, 1024 = per-row decision dump into lse2, 4096 = s_memtime phase stamps of workgroup 0 behind lse2 (tools/probes/
// attn_pipe_stamps.py).  The product is always built with 0.
#ifndef FS2_ATTN_PROBE
#define FS2_ATTN_PROBE 0
#endif
constexpr int kProbe = FS2_ATTN_PROBE;
constexpr int kD = 128;          // head dim
constexpr int kRB = 256;         // bytes per K / V row in LDS
constexpr int kTileB = 16384;    // 64 keys x 256 B
constexpr float kThr = 6.0f;     // deferred-rescale threshold, log2 units: p <= 64, bf16 keeps its 8 bits at any scale

typedef __attribute__((ext_vector_type(4))) short tr_s4_t;
__device__ inline uint2 tr_read64(const unsigned char* lds_addr) {  // lane i of a 16-group receives column i of a 4 x 16 block
    const tr_s4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s4_t*)lds_addr);
    return *(const uint2*)&v;
}
__device__ inline float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ inline bf16x8_t as_bf(const u32x4_t& u) { return *(const bf16x8_t*)&u; }
// The two products.  ACC = false: the builtin (256-register kernel, everything in arch VGPRs).  ACC = true (64 queries per wave,
// one wave per SIMD, 512 registers): hipcc, once a kernel may use AGPRs, selects the AGPR form for EVERY builtin MFMA - the score
// accumulators the softmax reads with VALU instructions included, i.e. a v_accvgpr_read per element - so the MFMAs are asm
// statements that name the register file per operand: scores and their C operand in VGPRs, Q (B operand) and O (C / D) in the
// accumulator file, where nothing but the MFMAs and the rare rescale touches them (cdna_hip_programming.md §5.7).  hipcc pads no
// hazards around asm: the callers keep >= 2 MFMAs between an MFMA's D and its first VALU reader (fences below), and the rare
// paths that touch O or minit carry their own s_nop.
template <bool ACC>
__device__ __forceinline__ void mma_qk_first(f32x16_t& d, const u32x4_t& k, const u32x4_t& q, const f32x16_t& c) {
    if constexpr (ACC) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c));
    } else {
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(k), as_bf(q), c, 0, 0, 0);
    }
}
template <bool ACC>
__device__ __forceinline__ void mma_qk(f32x16_t& d, const u32x4_t& k, const u32x4_t& q) {
    if constexpr (ACC) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
    } else {
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(k), as_bf(q), d, 0, 0, 0);
    }
}
// 96 queries per wave (NQB = 3): no reference tuple at all (the chain starts from the inline constant 0), and the third block's Q
// fragments come from LDS into arch VGPRs (the accumulator file holds O of three blocks and Q of two: 256 registers)
__device__ __forceinline__ void mma_qk_first0(f32x16_t& d, const u32x4_t& k, const u32x4_t& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "a"(q));
}
__device__ __forceinline__ void mma_qk_first0_v(f32x16_t& d, const u32x4_t& k, const u32x4_t& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "v"(q));
}
__device__ __forceinline__ void mma_qk_v(f32x16_t& d, const u32x4_t& k, const u32x4_t& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "v"(q));
}
template <bool ACC>
__device__ __forceinline__ void mma_pv(f32x16_t& o, const u32x4_t& v, const u32x4_t& pw) {
    if constexpr (ACC) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pw));
    } else {
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(v), as_bf(pw), o, 0, 0, 0);
    }
}

// Per-wave state of one item (NQB query blocks of 32)
template <int NQB>
struct PipeState {
    f32x16_t S0[NQB], S1[NQB];   // two live half tiles of scores (minus the reference max), log2 units
    f32x16_t minit[NQB];         // -(reference max), sixteen copies: the C operand that starts a Q.K^T chain
    f32x16_t oacc[NQB][4];       // O^T: dv block nd, lane = query
    uint32_t P0[NQB][8], P1[NQB][8];  // P of the two half tiles in flight, bf16 pairs: words 0-3 = 16-key chunk 0, 4-7 = chunk 1
    float lsum[NQB][2];          // this lane's share of the denominator (its 16 keys of every 32), two chains
    float mref[NQB];             // the reference the exponentials refer to: 0 in the first pass
    float mnext[NQB];            // a row whose sums ran over: the reference its next pass starts from
};

// No running max, and in the first pass no reference at all: p = exp2(s) with s the scaled score in log2 units (reference 0).
// bf16 keeps its 8 bits at any scale and O / the denominator accumulate in fp32, so the scale of a row's scores costs no accuracy
// - only a row whose denominator leaves [kLow, kBound] (scores beyond +-100 log2 units = 69 nats, inf, NaN) is wrong, and that row
// alone runs again from the reference log2(denominator) (the pass loop in run_item), which then rides in the C operand of the first
// MFMA of every chain.  The per-element max, the rescale decision and its branch were a fifth of the loop's issue slots and the
// only VALU use of the output accumulators; the first half tile's row max (the first version's reference) was ~1.5 k unhidden
// cycles per item.
constexpr float kBound = 0x1p100f, kLow = 0x1p-100f;

template <int NQB, bool ACC>
__device__ __forceinline__ bool run_item(const AttnArgs& p, unsigned char* smem, int bh, int q0, int wave, int lane) {
    static_assert(NQB != 3 || ACC, "96 queries per wave need the whole register file");
    constexpr int NQR = NQB == 3 ? 2 : NQB;  // query blocks whose Q fragments stay in registers
    const int li = lane & 31, hi = lane >> 5;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int ld = 3 * p.H;
    const bf16* __restrict__ qkv = (const bf16*)p.qkv;
    // valid-key words of this utterance: lane w holds word w (<= 64 tiles: S <= 4096), a tile's word is read back with
    // v_readlane (uniform loads of memory this launch also writes are not scalarised by hipcc; a vector load per tile would put
    // a vmcnt(0) behind the tile DMAs)
    const int ntiles = (p.S + 63) >> 6;
    unsigned long long myword = 0ull;
    if (lane < ntiles) myword = p.kbits[(size_t)b * p.nw64 + lane];
    const unsigned wlo = (unsigned)myword, whi = (unsigned)(myword >> 32);
    const unsigned long long nz = __ballot(myword != 0ull);
    const int jb = nz ? __builtin_ctzll(nz) : 0;
    const int je = nz ? 64 - __builtin_clzll(nz) : 0;
    auto tile_lo = [&](int t) { return (unsigned)__builtin_amdgcn_readlane((int)wlo, t); };
    auto tile_hi = [&](int t) { return (unsigned)__builtin_amdgcn_readlane((int)whi, t); };

    bf16* const outp = (bf16*)p.out;
    int qrow[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) qrow[qb] = q0 + (wave * NQB + qb) * 32 + li;

    if (jb == je) {  // every key padded: the reference's softmax over all -inf gives NaN
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
            if (qrow[qb] < p.S) {
                bf16 nanv; nanv.v = 0x7fc0;
                bf16* dst = outp + (size_t)(b * p.S + qrow[qb]) * p.H + h * kD + hi * 64;
                for (int e = 0; e < 64; ++e) dst[e] = nanv;
                if (p.lse2 && hi == 0) p.lse2[(size_t)bh * p.S + qrow[qb]] = __builtin_nanf("");
            }
        return false;
    }

    // smem: two V slots (0, 16 KiB), then three K slots (32, 48, 64 KiB)

    // ---- Q fragments first (they return to registers; the DMAs behind them are counted separately) ----
    uint4 qraw[NQB][8];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int r = qrow[qb] < p.S ? qrow[qb] : p.S - 1;
        const bf16* src = qkv + (size_t)(b * p.S + r) * ld + h * kD + hi * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) qraw[qb][c] = *(const uint4*)(src + c * 16);
    }

    // ---- K / V tile DMA: descriptor over this utterance's qkv rows; keys past its end read zeros (bounds check) ----
    const unsigned utt_bytes = (unsigned)((size_t)p.S * ld * 2);
    // piece i of a tile (16 per tile, 4 per wave) covers rows 16 i + 4 wave + (lane >> 4): the swizzle terms (row & 15, row & 3)
    // do not depend on i, so one per-lane offset serves all four pieces; the piece's row advance is a scalar
    unsigned kvo, vvo;
    {
        const int row = wave * 4 + (lane >> 4), ps = lane & 15;
        kvo = (unsigned)((row * ld + p.H + h * kD) * 2 + ((ps ^ (row & 15)) << 4));
        vvo = (unsigned)((row * ld + 2 * p.H + h * kD) * 2 + ((ps ^ ((row & 3) << 2)) << 4));
    }
    const unsigned ktile = (unsigned)(64 * ld * 2), kpiece = (unsigned)(16 * ld * 2);
    // The DMA is issued from an asm statement: hipcc's own scoreboard for buffer_load ... lds cannot tell the slots of one
    // LDS array apart and puts a vmcnt(0) in front of the next ds_read - i.e. right behind the issue.  Completion is counted
    // here instead: dma_drain() ahead of the one barrier per tile.
    typedef __attribute__((ext_vector_type(4))) int rsrc_words_t;
    const unsigned long long ubase = (unsigned long long)(uintptr_t)(qkv + (size_t)b * p.S * ld);
    rsrc_words_t qrw;  // raw buffer descriptor: base, stride 0, num_records = bytes, 32-bit untyped data format
    qrw[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ubase);
    qrw[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ubase >> 32) & 0xffffu));
    qrw[2] = __builtin_amdgcn_readfirstlane((int)utt_bytes);
    qrw[3] = 0x00020000;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 1024;
    auto issue_piece = [&](unsigned slot_off, unsigned vo, unsigned tile_off, int i) {
        const unsigned voff = vo + (tile_off + i * kpiece);
        const unsigned m0v = lds0 + slot_off + i * 4096;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(qrw) : "memory");
    };
    auto issue_tile = [&](unsigned slot_off, unsigned vo, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(slot_off, vo, (unsigned)t * ktile, i);
    };
    constexpr unsigned kK0 = 2 * kTileB;  // first K slot

    // ---- LDS fragment addresses ----
    // K fragment (key block kb, chunk c): row kb*32 + li, 16-byte slot (2c + hi) ^ (li & 15)  ==  kaddr ^ (c << 5), + kb * 8192
    // (chunks 4..7 = kaddr[c - 4] ^ 128, formed at the read); kaddr carries the base of the slot being read and moves by a scalar
    // delta when the loop turns to the next tile's slot (three slots: a tile's fragments are still being read while the DMA of
    // the tile after next is in flight, and a fragment ring of four registers replaces a phase's eight)
    unsigned kaddr[4];
    {
        const unsigned base0 = kK0 + (unsigned)(li * kRB + ((hi ^ (li & 15)) << 4));
#pragma unroll
        for (int c = 0; c < 4; ++c) kaddr[c] = base0 ^ (unsigned)(c << 5);
    }
    // V fragment (16-key chunk ch, dv block nd): two transposed 4 x 16 reads, 8 key rows apart; lane -> (key & 3, dv) as in
    // attention.hip: elements 0..3 <- keys ch*16 + 4hi + 0..3, 4..7 <- +8: the k-slot <-> key map the P registers carry
    unsigned vaddr[4];
    {
        const int i16 = lane & 15, g1 = (lane >> 4) & 1, rsub = i16 >> 2;
        const int rowb = (4 * hi + rsub) * kRB + (i16 & 1) * 8;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd)
            vaddr[nd] = (unsigned)(rowb + (((nd * 4 + g1 * 2 + ((i16 & 3) >> 1)) ^ (rsub << 2)) << 4));
    }

    PipeState<NQB> st;
    u32x4_t qf[NQR][8];
    // NQB = 3: the third block's eight fragments live in LDS behind the tile slots, lane-linear (a lane reads back exactly the 16
    // bytes it wrote: 1 KiB per fragment and wave, conflict-free), and pass through a four-deep register ring
    unsigned char* const q2base = smem + 5 * kTileB + wave * 8192 + lane * 16;
    u32x4_t q2[4];
    auto q2load = [&](int c) { return *(const u32x4_t*)(q2base + c * 1024); };
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {  // q * log2(e)/sqrt(d), once: scores leave the MFMA in exp2 units
            float f[8];
            Vec16<bf16>::unpack(qraw[qb][c], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2e;
            const uint4 pk = Vec16<bf16>::pack(f);
            if (qb < NQR) {
                qf[qb < NQR ? qb : 0][c] = u32x4_t{pk.x, pk.y, pk.z, pk.w};
                if (ACC) asm volatile("" : "+a"(qf[qb < NQR ? qb : 0][c]));  // one 128-bit accumulator-file tuple from here on (else: four scalars + copies per use)
            } else {
                *(uint4*)(q2base + c * 1024) = pk;
            }
        }
        st.mnext[qb] = 0.f;
    }
    bool done[NQB];         // this row's result has been stored by an earlier pass
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) done[qb] = false;
    unsigned ks_cur = kK0, ks_nxt = kK0 + kTileB, ks_fre = kK0 + 2 * kTileB;  // K(t), K(t+1), K(t+2)

    u32x4_t kf[4], vf[4];  // fragment rings: a K / V fragment is requested about eight MFMAs ahead of its use
    auto kload = [&](int kb, int c) { return *(const u32x4_t*)(smem + (c < 4 ? kaddr[c] : (kaddr[c - 4] ^ 128u)) + kb * 8192); };
    auto vload = [&](int ch, int nd) {
        const unsigned char* vb = smem + vaddr[nd] + ch * (16 * kRB);
        const uint2 lo = tr_read64(vb);
        const uint2 hi2 = tr_read64(vb + 8 * kRB);
        return u32x4_t{lo.x, lo.y, hi2.x, hi2.y};
    };
    // unit u of a half tile (U = 8 NQB of them): pair k = u / NQB of query block u % NQB: accumulator registers 2k, 2k+1 ->
    // exp2, row sums, one packed bf16 word.  Five instructions: what one MFMA hides (MI355X_MICROARCH.md, one wave per SIMD)
    auto unit = [&](f32x16_t (&S)[NQB], uint32_t (&P)[NQB][8], int u) {
        if (kProbe & 1) return;
        const int k = u / NQB, qb = u % NQB;
        const float e0 = __builtin_amdgcn_exp2f(S[qb][2 * k]);
        const float e1 = __builtin_amdgcn_exp2f(S[qb][2 * k + 1]);
        st.lsum[qb][0] += e0;
        st.lsum[qb][1] += e1;
        P[qb][k] = pack_bf16x2(e0, e1);
        // anchor: without it the optimiser sinks the whole slice below the phase (its results are used a phase later)
        asm volatile("" : "+v"(P[qb][k]), "+v"(st.lsum[qb][0]), "+v"(st.lsum[qb][1]));
    };
    auto mask_half = [&](f32x16_t (&S)[NQB], unsigned bits32) {  // keys of this 32-key block that are padded -> -inf
        if (ACC) {  // S left the matrix pipe just now: MFMA D -> VALU write needs the wait states hipcc does not insert for asm
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(S[qb]));
        }
        const unsigned w = bits32 >> (4 * hi);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (!((w >> ((r & 3) + 8 * (r >> 2))) & 1u)) S[qb][r] = -INFINITY;
    };
    auto qk_mma = [&](f32x16_t (&Sn)[NQB], int i, int qb) {  // chunk i of Q.K^T for query block qb; K fragment in the ring
        if constexpr (NQB == 3) {
            if (qb < 2) {
                if (i == 0) mma_qk_first0(Sn[qb], kf[0], qf[qb < 2 ? qb : 0][0]);
                else mma_qk<true>(Sn[qb], kf[i & 3], qf[qb < 2 ? qb : 0][i]);
            } else {
                if (i == 0) mma_qk_first0_v(Sn[qb], kf[0], q2[0]);
                else mma_qk_v(Sn[qb], kf[i & 3], q2[i & 3]);
            }
        } else {
            if (i == 0) mma_qk_first<ACC>(Sn[qb], kf[0], qf[qb < NQR ? qb : 0][0], st.minit[qb]);
            else mma_qk<ACC>(Sn[qb], kf[i & 3], qf[qb < NQR ? qb : 0][i]);
        }
    };
    constexpr int G = 8 * NQB;      // MFMAs (= gaps) per phase
    constexpr int UH = 4 * NQB;     // units per half of a half tile's finishing
    // Q.K^T of the next half tile into Sn || second half of the units of Sc -> Pc || K ring refill (fragments 4..7 of block kb
    // of the tile kaddr points at) || the first four V fragments of the coming P.V phase (chunk vch) || tile DMA pieces (hook).
    // One MFMA, then at most about five other instructions, then the next MFMA: an in-order wave gets VALU issue slots only in
    // the shadow of ITS OWN last MFMA - two MFMAs back to back waste the first one's (MI355X_MICROARCH.md, "two waves per SIMD" 1).
    auto phase_qk = [&](f32x16_t (&Sn)[NQB], f32x16_t (&Sc)[NQB], uint32_t (&Pc)[NQB][8], int kb, int vch, auto DO_MMA, auto&& hook) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = g / NQB, qb = g % NQB;
            if (decltype(DO_MMA)::value && !(kProbe & 16)) qk_mma(Sn, i, qb);
            if (g & 1) {
                unit(Sc, Pc, UH + (g >> 1));
            } else if (NQB == 3) {
                // twelve even gaps: chunk j's last MFMA (block 2) is gap 3j + 2 - its K ring slot is refilled in the next even gap,
                // its Q ring slot two gaps on; the ring's chunks 0, 1 were requested at the end of the P.V phase before
                const int e = g >> 1;  // 0..11
                constexpr bool mm = decltype(DO_MMA)::value;
                if (!(kProbe & 4)) {
                    if (mm && e == 0) q2[2] = q2load(2);
                    if (mm && e == 1) kf[0] = kload(kb, 4);
                    if (mm && e == 2) q2[3] = q2load(3);
                    if (mm && e == 3) { kf[1] = kload(kb, 5); q2[0] = q2load(4); }
                    if (mm && e == 4) { kf[2] = kload(kb, 6); q2[1] = q2load(5); }
                    if (mm && e == 5) q2[2] = q2load(6);
                    if (mm && e == 6) { kf[3] = kload(kb, 7); q2[3] = q2load(7); }
                    if (e >= 7 && e <= 10) vf[e - 7] = vload(vch, e - 7);
                }
                if (e == 1) hook(0);
                if (e == 2) hook(1);
                if (e == 5) hook(2);
                if (e == 7) hook(3);
            } else if (NQB == 2) {
                const int e = g >> 1;  // 0..7
                if (e == 0 && !(kProbe & 4)) vf[0] = vload(vch, 0);
                if (e >= 1 && e <= 4) {
                    if (decltype(DO_MMA)::value && !(kProbe & 4)) kf[e - 1] = kload(kb, e + 3);
                    hook(e - 1);
                }
                if (e >= 5 && !(kProbe & 4)) vf[e - 4] = vload(vch, e - 4);
            } else {
                const int e = g >> 1;  // 0..3
                if (decltype(DO_MMA)::value && !(kProbe & 4)) {
                    if (e == 0) kf[0] = kload(kb, 4);
                    if (e == 1) { kf[1] = kload(kb, 5); kf[2] = kload(kb, 6); }
                    if (e == 2) kf[3] = kload(kb, 7);
                }
                if (!(kProbe & 4)) vf[e] = vload(vch, e);
                hook(e);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // P.V of the finished half tile Pc (16-key chunks ch0, ch0 + 1) || first half of the units of Sn -> Pn || V ring refill (chunk
    // ch0 + 1) || fragments 0..3 of key block kbn of the tile kaddr points at, for the coming Q.K^T phase || tile DMA pieces
    auto phase_pv = [&](uint32_t (&Pc)[NQB][8], f32x16_t (&Sn)[NQB], uint32_t (&Pn)[NQB][8], int ch0, int kbn, auto DO_UNITS, auto DO_KLOAD,
                        auto&& hook) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = g / NQB, qb = g % NQB;
            {
                const u32x4_t pf = (i >> 2) == 0 ? u32x4_t{Pc[qb][0], Pc[qb][1], Pc[qb][2], Pc[qb][3]}
                                                 : u32x4_t{Pc[qb][4], Pc[qb][5], Pc[qb][6], Pc[qb][7]};
                if (!(kProbe & 32)) mma_pv<ACC>(st.oacc[qb][i & 3], vf[i & 3], pf);
            }
            if (g & 1) {
                // the scores these units read left the matrix pipe with the LAST MFMAs of the phase before; the first unit (query
                // block 0) sits behind three later MFMAs, the second (block NQB - 1) behind four or more: asm statements keep order
                if (decltype(DO_UNITS)::value) unit(Sn, Pn, g >> 1);
            } else if (NQB == 3) {
                const int e = g >> 1;  // 0..11; step j's MFMAs are gaps 3j .. 3j + 2: its V ring slot is refilled from gap 3j + 3 on
                constexpr bool kl = decltype(DO_KLOAD)::value;
                if (!(kProbe & 4)) {
                    if (kl && e == 0) kf[0] = kload(kbn, 0);
                    if (kl && e == 1) kf[1] = kload(kbn, 1);
                    if (e == 2) vf[0] = vload(ch0 + 1, 0);
                    if (e == 3) vf[1] = vload(ch0 + 1, 1);
                    if (e == 5) vf[2] = vload(ch0 + 1, 2);
                    if (e == 6) vf[3] = vload(ch0 + 1, 3);
                    if (kl && e == 7) kf[2] = kload(kbn, 2);
                    if (kl && e == 8) kf[3] = kload(kbn, 3);
                    if (kl && e == 10) q2[0] = q2load(0);  // the coming Q.K^T phase's first two fragments of block 2
                    if (kl && e == 11) q2[1] = q2load(1);
                }
                if (e == 0) hook(0);
                if (e == 4) hook(1);
                if (e == 7) hook(2);
                if (e == 9) hook(3);
            } else if (NQB == 2) {
                const int e = g >> 1;  // 0..7
                if (e >= 1 && e <= 4 && !(kProbe & 4)) vf[e - 1] = vload(ch0 + 1, e - 1);
                if (e == 0 || e >= 5) {
                    const int j = e == 0 ? 0 : e - 4;
                    if (decltype(DO_KLOAD)::value && !(kProbe & 4)) kf[j] = kload(kbn, j);
                    hook(j);
                }
            } else {
                const int e = g >> 1;  // 0..3
                if (!(kProbe & 4)) {
                    if (e == 1) { vf[0] = vload(ch0 + 1, 0); vf[1] = vload(ch0 + 1, 1); }
                    if (e == 2) { vf[2] = vload(ch0 + 1, 2); vf[3] = vload(ch0 + 1, 3); }
                    if (decltype(DO_KLOAD)::value) {
                        if (e == 0) kf[0] = kload(kbn, 0);
                        if (e == 3) { kf[1] = kload(kbn, 1); kf[2] = kload(kbn, 2); kf[3] = kload(kbn, 3); }
                    }
                }
                hook(e);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // One tile of the main loop: one barrier per 64-key tile
    auto tile_step = [&](int t, auto HN, auto MK) {
        constexpr bool has_next = decltype(HN)::value, masked = decltype(MK)::value && !(kProbe & 512);
        const int rel = t - jb;
        if (!(kProbe & 256)) {
            dma_drain();
            __syncthreads();  // V(t), K(t+1) landed for everyone; K(t-1) and V(t-1) are dead
        }
        auto no_hook = [](int) {};
        // K(t+2) rides in phase C, V(t+1) in phase D, a piece per even gap.  No branch in a gap: a K tile past the end is issued
        // with an offset beyond the descriptor's range (zeros land in a slot nobody reads again)
        constexpr bool dma = has_next && !(kProbe & 8);
        const unsigned koff = t + 2 < je ? (unsigned)(t + 2) * ktile : 0x80000000u, voff = (unsigned)(t + 1) * ktile;
        const unsigned vslot = ((rel + 1) & 1) * kTileB;
        auto hook_k = [&](int i) { if (dma) issue_piece(ks_fre, kvo, koff, i); };
        auto hook_v = [&](int i) { if (dma) issue_piece(vslot, vvo, voff, i); };

        // C: Q.K^T(t, keys 32..63) -> S1 || units of S0, second half || V(t) chunk 0 || K(t) block 1, fragments 4..7
        phase_qk(st.S1, st.S0, st.P0, 1, 0, std::true_type{}, hook_k);
        if (masked) {
            const unsigned bits = tile_hi(t);
            if (bits != 0xffffffffu) mask_half(st.S1, bits);
        }
        if (has_next) {  // fragment reads turn to K(t+1)'s slot
            const unsigned dk = ks_nxt - ks_cur;
#pragma unroll
            for (int c = 0; c < 4; ++c) kaddr[c] += dk;
        }
        // D: P.V(t, keys 0..31) || units of S1, first half || V(t) chunk 1 || K(t+1) block 0, fragments 0..3
        phase_pv(st.P0, st.S1, st.P1, 0, 0, std::true_type{}, HN, hook_v);
        // A': Q.K^T(t+1, keys 0..31) -> S0 || units of S1, second half || V(t) chunk 2 || K(t+1) block 0, fragments 4..7
        phase_qk(st.S0, st.S1, st.P1, 0, 2, HN, no_hook);
        if (has_next && masked) {
            const unsigned bitsn = tile_lo(t + 1);
            if (bitsn != 0xffffffffu) mask_half(st.S0, bitsn);
        }
        // B': P.V(t, keys 32..63) || units of S0, first half || V(t) chunk 3 || K(t+1) block 1, fragments 0..3
        phase_pv(st.P1, st.S0, st.P0, 2, 1, HN, HN, no_hook);
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) vaddr[nd] ^= (unsigned)kTileB;
        if (has_next) {
            const unsigned tmp = ks_cur;
            ks_cur = ks_nxt; ks_nxt = ks_fre; ks_fre = tmp;
        }
    };

    // tiles jb .. jb + nfull - 1 have all 64 keys valid: their steps carry no mask test (a test is a v_readlane, a compare and a
    // branch that ends the scheduling region - ~50 cycles each, twice per tile)
    int nfull;
    {
        const unsigned long long notfull = __ballot(myword != ~0ull) >> jb;
        nfull = notfull ? __builtin_ctzll(notfull) : 64 - jb;
    }
    constexpr int kMaxPass = NQB == 3 ? 1 : 64;  // 96 queries per wave: one pass, the caller reruns the units on the paths that carry a reference

    long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) { if (kProbe & 4096) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); };
    // ---- passes over the item: one, unless a row's sums ran over kBound; bounded (NaN / inf scores never settle) ----
    for (int pass = 0; pass < kMaxPass; ++pass) {
        stamp(0);
        __syncthreads();  // every wave is done with the previous item's / pass's tiles (and has read its flags)
        {   // fragment addresses back to the first slots
            const unsigned dk = kK0 - ks_cur;
#pragma unroll
            for (int c = 0; c < 4; ++c) kaddr[c] += dk;
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) vaddr[nd] &= ~(unsigned)kTileB;
            ks_cur = kK0; ks_nxt = kK0 + kTileB; ks_fre = kK0 + 2 * kTileB;
        }
        issue_tile(kK0, kvo, jb);
        issue_tile(0, vvo, jb);
        if (jb + 1 < je) issue_tile(kK0 + kTileB, kvo, jb + 1);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            st.mref[qb] = pass == 0 ? 0.f : st.mnext[qb];
#pragma unroll
            for (int r = 0; r < 16; ++r) st.minit[qb][r] = -st.mref[qb];
#pragma unroll
            for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                for (int r = 0; r < 16; ++r) st.oacc[qb][nd][r] = 0.f;
            if (ACC) {
                asm volatile("" : "+v"(st.minit[qb]));
#pragma unroll
                for (int nd = 0; nd < 4; ++nd) asm volatile("" : "+a"(st.oacc[qb][nd]));
            }
            st.lsum[qb][0] = st.lsum[qb][1] = 0.f;
        }

        // ---- prologue: scores of the first half tile of tile jb, the first half of their units ----
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // Q and K(jb): everything but the two youngest tile DMAs (4 pieces each)
        if (jb + 1 >= je) dma_drain();                    // (only two tiles were issued: wait for both)
        __syncthreads();
        stamp(1);
        {   // scores of the first half tile (block 0 of tile jb): nothing to overlap with
#pragma unroll
            for (int c = 0; c < 4; ++c) kf[c] = kload(0, c);
            if (NQB == 3) {
#pragma unroll
                for (int c = 0; c < 4; ++c) q2[c] = q2load(c);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) qk_mma(st.S0, i, qb);
                // this block's fragments 4..7, then block 1's 0..3 for the first phase C
                kf[i & 3] = kload(i < 4 ? 0 : 1, i < 4 ? i + 4 : i - 4);
                if (NQB == 3) q2[i & 3] = q2load(i < 4 ? i + 4 : i - 4);  // chunks 4..7, then 0..3 again for phase C (its first two are used)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ACC) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // MFMA D -> VALU read (prologue only)
            const unsigned bits = tile_lo(jb);
            if (bits != 0xffffffffu) mask_half(st.S0, bits);
        }
#pragma unroll
        for (int u = 0; u < UH; ++u) unit(st.S0, st.P0, u);

        stamp(2);
        // ---- main loop ----
        {
            int t = jb;
            const int tu = jb + nfull - 1 < je - 1 ? jb + nfull - 1 : je - 1;
            for (; t < tu; ++t) tile_step(t, std::true_type{}, std::false_type{});
            for (; t + 1 < je; ++t) tile_step(t, std::true_type{}, std::true_type{});
            tile_step(je - 1, std::false_type{}, std::true_type{});
        }

        // ---- normalise and store: lane (li, hi) owns query li, dv = nd*32 + 8g + 4hi + 0..3; the two halves of a row trade
        // 8-byte groups (v_permlane32_swap) so that every lane stores 16 contiguous bytes ----
        stamp(3);
        if (ACC) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // last P.V MFMAs -> O read
        bool more = false;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const float l = xhalf_sum(st.lsum[qb][0] + st.lsum[qb][1]);
            const float inv = 1.f / l;
            const bool ok = l <= kBound && l >= kLow;  // false for inf and NaN too
            const bool mine = qrow[qb] < p.S && !done[qb] && (ok || (NQB != 3 && pass == kMaxPass - 1));
            // a row that ran over (under) starts again from about its largest score: log2(l) is within log2(S) of (max - reference);
            // an infinite (zero) denominator moves the reference by a fixed +-100 and tries again
            st.mnext[qb] = st.mref[qb] + (l < INFINITY ? (l > 0.f ? __builtin_amdgcn_logf(l) : -100.f) : 100.f);
            if ((kProbe & 1024) && p.lse2 && hi == 0 && qrow[qb] < p.S)  // debug dump: one record per (row, pass)
                ((float4*)p.lse2)[(size_t)qrow[qb] * 4 + (pass < 4 ? pass : 3)] = make_float4(l, st.mref[qb], ok ? 1.f : 0.f, (mine ? 1.f : 0.f) + (done[qb] ? 2.f : 0.f) + 10.f * pass);
            if (mine) {
                if (p.lse2 && hi == 0) p.lse2[(size_t)bh * p.S + qrow[qb]] = st.mref[qb] + __builtin_amdgcn_logf(l);  // v_log_f32 = log2
            }
            bf16* dst = outp + (size_t)(b * p.S + (qrow[qb] < p.S ? qrow[qb] : 0)) * p.H + h * kD + hi * 8;
            uint4 ov[8];  // all eight pieces first, then ONE predicated block of stores (a branch per store ends the scheduling region
                          // sixteen times per block; staging the rows through LDS for whole-row stores measured no faster, r03)
#pragma unroll
            for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const f32x16_t& o = st.oacc[qb][nd];
                    uint32_t a0 = pack_bf16x2(o[4 * g + 0] * inv, o[4 * g + 1] * inv), a1 = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                    uint32_t b0 = pack_bf16x2(o[4 * g + 4] * inv, o[4 * g + 5] * inv), b1 = pack_bf16x2(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    // lower half: [own g | upper's g] = dv 8g .. 8g+7; upper half: [lower's g+1 | own g+1] = dv 8g+8 .. 8g+15
                    ov[nd * 2 + (g >> 1)] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
            if (mine) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *(uint4*)(dst + (i >> 1) * 32 + 16 * (i & 1)) = ov[i];
            }
            done[qb] = done[qb] || ok;
            more = more || !done[qb];
        }
        // does any row of the workgroup need another pass?  The waves trade a word each through a K slot nobody has read or
        // written since the barrier of the last tile (K(t+1)'s: the last tile has no next); the barrier at the top of the next
        // pass / item keeps the next DMA away from it until every wave has read
        stamp(4);
        if ((kProbe & 4096) && p.lse2 && blockIdx.x == 0 && wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(5);
            if (lane == 0) {
                float* o = p.lse2 + (size_t)p.B * p.heads * p.S + (size_t)(q0 / 128) * 8;   // one record per unit index of this workgroup's items
                for (int i = 0; i < 6; ++i) o[i] = (float)(stamps[i] - stamps[0]);
                o[6] = (float)NQB; o[7] = (float)(stamps[0] & 0xffffff);
            }
        }
        {
            int* flags = (int*)(smem + ks_nxt);
            const int wave_more = __any(more) ? 1 : 0;
            if (lane == 0) flags[wave] = wave_more;
            __syncthreads();
            const int4 f = *(const int4*)flags;
            const int again = __builtin_amdgcn_readfirstlane(f.x | f.y | f.z | f.w);
            if (NQB == 3) return again != 0;
            if (!again) break;
        }
    }
    return false;
}

template <int NQBMAX>
__global__ __launch_bounds__(256, NQBMAX >= 2 ? 1 : 2) void attention_pipe_kernel(AttnArgs p, int nu, int U) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(32768))) unsigned char smem[5 * kTileB + (NQBMAX == 3 ? 32768 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;  // a multiple of 8: slot = XCD-major so that an XCD's workgroups own neighbouring units
    const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    int u = (int)((long long)slot * U / G);
    const int u1 = (int)((long long)(slot + 1) * U / G);
    while (u < u1) {
        const int bh = u / nu, uu = u - bh * nu;
        if constexpr (NQBMAX == 3) {
            if (!(kProbe & 128) && uu + 3 <= nu && u + 3 <= u1) {
                // 384 queries, 96 per wave, one pass; a row outside the reference-free range sends the three units to the paths below
                if (run_item<3, true>(p, smem, bh, uu * 128, wave, lane)) {
                    run_item<2, true>(p, smem, bh, uu * 128, wave, lane);
                    run_item<1, true>(p, smem, bh, (uu + 2) * 128, wave, lane);
                }
                u += 3;
                continue;
            }
        }
        if (NQBMAX >= 2 && !(kProbe & 128) && uu + 2 <= nu && u + 2 <= u1) {
            run_item<2, true>(p, smem, bh, uu * 128, wave, lane);
            u += 2;
        } else {
            run_item<1, NQBMAX >= 2>(p, smem, bh, uu * 128, wave, lane);
            u += 1;
        }
    }
#else
    (void)p; (void)nu; (void)U;
#endif
}

}  // namespace

// bf16, head dim 128, no attention-weight dropout; variant: 2 = 64 queries per wave (one workgroup per CU), 1 = 32 (two)
bool attention_pipe_supported(const AttnArgs& a, int dtype) {
    return dtype == FS2_BF16 && a.H == a.heads * kD && a.drop_p == 0.f && a.S >= 1 && a.S <= 4096;
}

int launch_attention_pipe(const AttnArgs& a, int variant, hipStream_t stream) {
    const int nu = (a.S + 127) / 128;
    const long long U = (long long)a.B * a.heads * nu;
    if (U > 0x7fffffffll) return FS2_ERR_SHAPE;
    if (variant == 3) {
        hipLaunchKernelGGL((attention_pipe_kernel<3>), dim3(256), dim3(256), 0, stream, a, nu, (int)U);
    } else if (variant == 2) {
        hipLaunchKernelGGL((attention_pipe_kernel<2>), dim3(256), dim3(256), 0, stream, a, nu, (int)U);
    } else {
        hipLaunchKernelGGL((attention_pipe_kernel<1>), dim3(512), dim3(256), 0, stream, a, nu, (int)U);
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2

#if FS2_ATTN_PROBE
extern "C" int attn_pipe_probe_dbg(const void* qkv, const uint64_t* kbits, void* out, float* dbg, int B, int S, int H, int heads, int variant, void* stream) {
    fs2::AttnArgs a;
    a.qkv = qkv; a.vt = nullptr; a.kbits = kbits; a.out = out; a.lse2 = dbg;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.Spad = (S + 63) / 64 * 64; a.nw64 = a.Spad / 64;
    a.scale_log2e = 1.4426950408889634f / 11.313708f;
    return fs2::launch_attention_pipe(a, variant, (hipStream_t)stream);
}
extern "C" int attn_pipe_probe(const void* qkv, const uint64_t* kbits, void* out, int B, int S, int H, int heads, int variant, void* stream) {
    fs2::AttnArgs a;
    a.qkv = qkv; a.vt = nullptr; a.kbits = kbits; a.out = out;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.Spad = (S + 63) / 64 * 64; a.nw64 = a.Spad / 64;
    a.scale_log2e = 1.4426950408889634f / 11.313708f;
    return fs2::launch_attention_pipe(a, variant, (hipStream_t)stream);
}
#endif
