// Software-pipelined fused attention for gfx950, bf16, head dim 128: the MFMA-bound instance of the attention inside
// nn.MultiheadAttention (ConformerEncoderLayer.forward, /root/reference/litfass/fastspeech2/model.py:108-116) - the decoder's
// self-attention over T frames.  Same arithmetic contract as attention.hip (padded KEYS masked, padded queries computed,
// base-2 online softmax with a deferred rescale, fp32 statistics), different schedule:
//
//   attention.hip runs a KV tile as  [16 Q.K^T MFMAs] [softmax VALU] [16 P.V MFMAs]  per wave: the matrix pipe idles while the
//   ~200 softmax instructions issue and the issue port idles while MFMAs drain (PMC: MFMA pipe 44 % busy, 5 VALU per MFMA), and
//   a second wave on the SIMD does not fill the holes (the arbiter serves the oldest wave).  Here every MFMA phase
//   carries the softmax of ANOTHER 32-key half tile of the same wave between its MFMAs:
//
//     phase QK(h+1):  S(h+1) = K(h+1) Q^T  [8 K fragments x NQB MFMAs]  ||  finish(h): p = exp2(S(h)), row sums, bf16 pack
//                                                                        ||  V(h) fragments -> registers (ds_read_b64_tr_b16)
//     phase PV(h):    O^T += V(h)^T P(h)^T [8 V fragments x NQB MFMAs]  ||  start(h+1): row max of S(h+1)
//                                                                        ||  K(h+2) fragments -> registers (ds_read_b128)
//     then the (rare) rescale decision for S(h+1), after every P.V MFMA of the phase has been issued (a rescale of O may not
//     split a pending P.V: cdna_hip_programming.md T13).
//
//   A wave owns NQB 32-query blocks that share every K / V fragment it reads (NQB = 2: half the LDS reads per MFMA); two S
//   half tiles (16 registers per block each) are alive at a time; the running max rides in a 16-register block that is the C
//   operand of the first Q.K^T MFMA of a chain, so p = exp2(acc) needs no subtract and no accumulator initialisation.
//   K / V tiles (64 keys) stream by buffer-load-to-LDS DMA into two slots each: ONE barrier per tile - at the top of tile t
//   everything tile t reads (V(t), K(t+1)) has landed, K(t) / V(t-1) are dead and K(t+2) / V(t+1) are issued into their slots, a
//   whole tile ahead of their first read.
//
// Work split: a unit is 128 queries of one (utterance, head); the grid is one workgroup per CU (NQB = 2, 512 registers, one wave
// per SIMD) or two (NQB = 1), each owning a contiguous run of units - pairs of units of one head run as 256-query items (64 per
// wave), leftovers as 128-query items - so 64 x 1536 queries (C2 decoder) are 3 units per workgroup, one round, no tail.
// Workgroups of one XCD own neighbouring units: a head's K / V stay in that XCD's L2.
#include "fs2_common.h"
#include "fs2_kernels.h"
#include <type_traits>

namespace fs2 {
namespace {

// Probe builds (tools/probes/attn_pipe_probe.py compiles this file with -DFS2_ATTN_PROBE=bits into throw-away libraries and times
// them): 1 = no exp / sum / pack, 2 = no row max / decision, 4 = no fragment reads in the loop, 8 = no tile DMA in the loop,
// 16 = no Q.K^T MFMAs, 32 = no P.V MFMAs.  The product is always built with 0.
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
    f32x16_t S0[NQB], S1[NQB];   // two live half tiles of scores (minus the running max), log2 units
    f32x16_t minit[NQB];         // -(running max), sixteen copies: the C operand that starts a Q.K^T chain
    f32x16_t oacc[NQB][4];       // O^T: dv block nd, lane = query
    uint32_t pfw[NQB][8];        // P of the half tile in flight, bf16 pairs: words 0-3 = 16-key chunk 0, 4-7 = chunk 1
    float lsum[NQB][2];          // this lane's share of the denominator (its 16 keys of every 32), two chains
    float thr[NQB];              // -inf until the row has a finite max, then kThr
    float mref[NQB];             // the running max the exponentials refer to
};

template <int NQB, bool ACC>
__device__ __forceinline__ void run_item(const AttnArgs& p, unsigned char* smem, int bh, int q0, int wave, int lane) {
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
        return;
    }

    // smem: two K slots, then two V slots
    __syncthreads();  // every wave is done with the previous item's tiles

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
    // The DMA is issued from an asm statement: hipcc's own scoreboard for buffer_load ... lds cannot tell the four slots of one
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
    auto issue_tile = [&](unsigned slot_off, unsigned vo, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned voff = vo + ((unsigned)t * ktile + i * kpiece);
            const unsigned m0v = lds0 + slot_off + i * 4096;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(qrw) : "memory");
        }
    };
    // one piece (1 KiB) of a tile: inside the loop the four pieces of a tile ride between the MFMAs of a phase, one per step - a
    // burst of eight behind the barrier keeps the matrix pipe idle for its whole issue time (~100 cycles a piece)
    auto issue_piece = [&](unsigned slot_off, unsigned vo, unsigned tile_off, int i) {
        const unsigned voff = vo + (tile_off + i * kpiece);
        const unsigned m0v = lds0 + slot_off + i * 4096;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(qrw) : "memory");
    };
    issue_tile(0, kvo, jb);
    issue_tile(2 * kTileB, vvo, jb);
    if (jb + 1 < je) issue_tile(kTileB, kvo, jb + 1);

    // ---- LDS fragment addresses (slot 0; XOR kTileB toggles the slot) ----
    // K fragment (key block kb, chunk c): row kb*32 + li, 16-byte slot (2c + hi) ^ (li & 15)  ==  kaddr ^ (c << 5), + kb * 8192
    // (chunks 4..7 = kaddr[c - 4] ^ 128, formed at the read: four registers less in a kernel that sits at the 256 cap)
    unsigned kaddr[4];
    {
        const unsigned base0 = (unsigned)(li * kRB + ((hi ^ (li & 15)) << 4));
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
            vaddr[nd] = (unsigned)(2 * kTileB + rowb + (((nd * 4 + g1 * 2 + ((i16 & 3) >> 1)) ^ (rsub << 2)) << 4));
    }

    PipeState<NQB> st;
    u32x4_t qf[NQB][8];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {  // q * log2(e)/sqrt(d), once: scores leave the MFMA in exp2 units
            float f[8];
            Vec16<bf16>::unpack(qraw[qb][c], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2e;
            const uint4 pk = Vec16<bf16>::pack(f);
            qf[qb][c] = u32x4_t{pk.x, pk.y, pk.z, pk.w};
            if (ACC) asm volatile("" : "+a"(qf[qb][c]));  // one 128-bit accumulator-file tuple from here on (else: four scalars + copies per use)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) st.minit[qb][r] = 0.f;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.oacc[qb][nd][r] = 0.f;
        st.lsum[qb][0] = st.lsum[qb][1] = 0.f;
        st.thr[qb] = -INFINITY;
        st.mref[qb] = 0.f;
    }

    u32x4_t kf[8], vf[4];  // K: a phase's eight fragments, read under the P.V phase before it; V: a four-deep ring
    auto kload = [&](int kb, int c) { return *(const u32x4_t*)(smem + (c < 4 ? kaddr[c] : (kaddr[c - 4] ^ 128u)) + kb * 8192); };
    auto vload = [&](int ch, int nd) {
        const unsigned char* vb = smem + vaddr[nd] + ch * (16 * kRB);
        const uint2 lo = tr_read64(vb);
        const uint2 hi2 = tr_read64(vb + 8 * kRB);
        return u32x4_t{lo.x, lo.y, hi2.x, hi2.y};
    };
    auto toggle_slots = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) kaddr[c] ^= (unsigned)kTileB;
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) vaddr[nd] ^= (unsigned)kTileB;
    };
    // pair k of a half tile: accumulator registers 2k, 2k+1 -> exp2, row sums, one packed bf16 word
    auto fin_pair = [&](f32x16_t (&S)[NQB], int k) {
        if (kProbe & 1) return;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const float e0 = __builtin_amdgcn_exp2f(S[qb][2 * k]);
            const float e1 = __builtin_amdgcn_exp2f(S[qb][2 * k + 1]);
            st.lsum[qb][0] += e0;
            st.lsum[qb][1] += e1;
            st.pfw[qb][k] = pack_bf16x2(e0, e1);
            // anchor: without it the optimiser sinks the whole slice below the phase (its results are used a phase later)
            asm volatile("" : "+v"(st.pfw[qb][k]), "+v"(st.lsum[qb][0]), "+v"(st.lsum[qb][1]));
        }
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
    // the decision for a freshly scored half tile whose row max (relative to the running max) is mx: only when a row's max grew
    // past the threshold (or the row has no finite max yet) is anything rescaled
    auto decide = [&](f32x16_t (&S)[NQB], float (&mx)[NQB]) {
        bool fire = false;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            mx[qb] = xhalf_max(mx[qb]);
            fire = fire || (mx[qb] > st.thr[qb]);
        }
        if (__any(fire)) {
            if (ACC) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // the phase's last P.V MFMAs -> O read below
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const bool f = mx[qb] > st.thr[qb];
                const float d = f ? mx[qb] : 0.f;
                const float alpha = st.thr[qb] == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(-d);
                st.thr[qb] = f ? kThr : st.thr[qb];
                st.mref[qb] += d;
                st.lsum[qb][0] *= alpha;
                st.lsum[qb][1] *= alpha;
#pragma unroll
                for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.oacc[qb][nd][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    S[qb][r] -= d;
                    st.minit[qb][r] = -st.mref[qb];
                }
            }
            if (ACC) {  // VALU / accvgpr writes -> MFMA operands of the next phase
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    asm volatile("s_nop 3" : "+v"(S[qb]), "+v"(st.minit[qb]));
#pragma unroll
                    for (int nd = 0; nd < 4; ++nd) asm volatile("" : "+a"(st.oacc[qb][nd]));
                }
            }
        }
    };

#ifndef FS2_ATTN_FQ
#define FS2_ATTN_FQ 6
#endif
    constexpr int FQ = FS2_ATTN_FQ;  // pairs of a half tile finished under the Q.K^T phase; the other 8 - FQ open the P.V phase

    // Q.K^T of the next half tile into Sn (K fragments in kf) || finish Sc || first four V fragments of the coming P.V phase
    auto phase_qk = [&](f32x16_t (&Sn)[NQB], f32x16_t (&Sc)[NQB], int ch_next, auto DO_MMA, auto&& step_hook) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (decltype(DO_MMA)::value && !(kProbe & 16)) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    if (i == 0) mma_qk_first<ACC>(Sn[qb], kf[i], qf[qb][i], st.minit[qb]);
                    else mma_qk<ACC>(Sn[qb], kf[i], qf[qb][i]);
                }
            }
            if (i < FQ) fin_pair(Sc, i);
            if (i >= 4 && !(kProbe & 4)) vf[i - 4] = vload(ch_next, i - 4);
            step_hook(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // P.V of the finished half tile (16-key chunks ch0, ch0 + 1) || row max of Sn || K fragments of key block kb_next -> kf
    auto phase_pv = [&](f32x16_t (&Sc)[NQB], f32x16_t (&Sn)[NQB], int ch0, int kb_next, auto DO_MAX, auto DO_KLOAD, float (&mx)[NQB],
                        auto&& step_hook) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 8 - FQ) fin_pair(Sc, FQ + i);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const u32x4_t pf = (i >> 2) == 0 ? u32x4_t{st.pfw[qb][0], st.pfw[qb][1], st.pfw[qb][2], st.pfw[qb][3]}
                                                 : u32x4_t{st.pfw[qb][4], st.pfw[qb][5], st.pfw[qb][6], st.pfw[qb][7]};
                if (!(kProbe & 32)) mma_pv<ACC>(st.oacc[qb][i & 3], vf[i & 3], pf);
            }
            if (ACC && i == 0 && decltype(DO_MAX)::value) {
                // the scores this phase takes the row max of left the matrix pipe with the LAST MFMAs of the phase before: no
                // VALU may read them until two more MFMAs (16 wait states) have been issued - asm statements keep their order
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) asm volatile("" : "+v"(Sn[qb]));
            }
            if (i < 4 && !(kProbe & 4)) vf[i] = vload(ch0 + 1, i);
            if (decltype(DO_MAX)::value && !(kProbe & 2)) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    mx[qb] = fmaxf(fmaxf(mx[qb], Sn[qb][2 * i]), Sn[qb][2 * i + 1]);
                    asm volatile("" : "+v"(mx[qb]));
                }
            }
            if (decltype(DO_KLOAD)::value && !(kProbe & 4)) kf[i] = kload(kb_next, i);
            step_hook(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: first half tile of tile jb ----
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // Q and K(jb): everything but the two youngest tile DMAs (4 pieces each)
    if (jb + 1 >= je) dma_drain();                    // (only two tiles were issued: wait for both)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i] = kload(0, i);
    {
        float mx[NQB];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                if (i == 0) mma_qk_first<ACC>(st.S0[qb], kf[i], qf[qb][i], st.minit[qb]);
                else mma_qk<ACC>(st.S0[qb], kf[i], qf[qb][i]);
            }
            kf[i] = kload(1, i);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ACC) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // MFMA D -> VALU read (prologue only)
        const unsigned bits = tile_lo(jb);
        if (bits != 0xffffffffu) mask_half(st.S0, bits);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            mx[qb] = st.S0[qb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx[qb] = fmaxf(mx[qb], st.S0[qb][r]);
        }
        decide(st.S0, mx);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) kaddr[c] ^= (unsigned)kTileB;  // kaddr -> slot of K(jb+1); vaddr stays on V(jb)'s

    // ---- main loop: one barrier per 64-key tile ----
    auto tile_step = [&](int t, auto HN) {
        constexpr bool has_next = decltype(HN)::value;
        const int rel = t - jb;
        dma_drain();
        __syncthreads();  // V(t), K(t+1) landed for everyone; K(t) and V(t-1) are dead
        auto no_hook = [](int) {};
        // K(t+2) rides in phase C, V(t+1) in phase D, a piece per step.  No branch in a step: a K tile past the end is issued
        // with an offset beyond the descriptor's range (zeros land in a slot nobody reads again)
        constexpr bool dma = has_next && !(kProbe & 8);
        const unsigned koff = t + 2 < je ? (unsigned)(t + 2) * ktile : 0x80000000u, voff = (unsigned)(t + 1) * ktile;
        const unsigned kslot = (rel & 1) * kTileB, vslot = (2 + ((rel + 1) & 1)) * kTileB;
        auto hook_k = [&](int i) { if (dma && i >= 4) issue_piece(kslot, kvo, koff, i - 4); };
        auto hook_v = [&](int i) { if (dma && i >= 2 && i < 6) issue_piece(vslot, vvo, voff, i - 2); };
        float mx[NQB];

        // C: Q.K^T(t, keys 32..63) || finish S0 || V(t) chunks 0, 1
        phase_qk(st.S1, st.S0, 0, std::true_type{}, hook_k);
        {
            const unsigned bits = tile_hi(t);
            if (bits != 0xffffffffu) mask_half(st.S1, bits);
        }
        // D: P.V(t, keys 0..31) || max S1 || K(t+1) keys 0..31
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) mx[qb] = -INFINITY;
        phase_pv(st.S0, st.S1, 0, 0, std::true_type{}, HN, mx, hook_v);
        if (!(kProbe & 2)) decide(st.S1, mx);
        // A': Q.K^T(t+1, keys 0..31) || finish S1 || V(t) chunks 2, 3
        phase_qk(st.S0, st.S1, 2, HN, no_hook);
        if (has_next) {
            const unsigned bitsn = tile_lo(t + 1);
            if (bitsn != 0xffffffffu) mask_half(st.S0, bitsn);
        }
        // B': P.V(t, keys 32..63) || max S0 || K(t+1) keys 32..63
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) mx[qb] = -INFINITY;
        phase_pv(st.S1, st.S0, 2, 1, HN, HN, mx, no_hook);
        if (has_next && !(kProbe & 2)) decide(st.S0, mx);
        toggle_slots();
    };
    for (int t = jb; t + 1 < je; ++t) tile_step(t, std::true_type{});
    tile_step(je - 1, std::false_type{});

    // ---- normalise and store: lane (li, hi) owns query li, dv = nd*32 + 8g + 4hi + 0..3; the two halves of a row trade
    // 8-byte groups (v_permlane32_swap) so that every lane stores 16 contiguous bytes ----
    if (ACC) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // last P.V MFMAs -> O read
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const float l = xhalf_sum(st.lsum[qb][0] + st.lsum[qb][1]);
        const float inv = 1.f / l;
        if (qrow[qb] < p.S) {
            if (p.lse2 && hi == 0) p.lse2[(size_t)bh * p.S + qrow[qb]] = st.mref[qb] + __builtin_amdgcn_logf(l);  // v_log_f32 = log2
        }
        bf16* dst = outp + (size_t)(b * p.S + (qrow[qb] < p.S ? qrow[qb] : 0)) * p.H + h * kD + hi * 8;
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
                if (qrow[qb] < p.S) *(uint4*)(dst + nd * 32 + 8 * g) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
    }
}

template <int NQBMAX>
__global__ __launch_bounds__(256, NQBMAX == 2 ? 1 : 2) void attention_pipe_kernel(AttnArgs p, int nu, int U) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(32768))) unsigned char smem[4 * kTileB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;  // a multiple of 8: slot = XCD-major so that an XCD's workgroups own neighbouring units
    const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    int u = (int)((long long)slot * U / G);
    const int u1 = (int)((long long)(slot + 1) * U / G);
    while (u < u1) {
        const int bh = u / nu, uu = u - bh * nu;
        if (NQBMAX == 2 && !(kProbe & 128) && uu + 2 <= nu && u + 2 <= u1) {
            run_item<NQBMAX, true>(p, smem, bh, uu * 128, wave, lane);
            u += 2;
        } else {
            run_item<1, NQBMAX == 2>(p, smem, bh, uu * 128, wave, lane);
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
    if (variant == 2) {
        hipLaunchKernelGGL((attention_pipe_kernel<2>), dim3(256), dim3(256), 0, stream, a, nu, (int)U);
    } else {
        hipLaunchKernelGGL((attention_pipe_kernel<1>), dim3(512), dim3(256), 0, stream, a, nu, (int)U);
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2

#if FS2_ATTN_PROBE
extern "C" int attn_pipe_probe(const void* qkv, const uint64_t* kbits, void* out, int B, int S, int H, int heads, int variant, void* stream) {
    fs2::AttnArgs a;
    a.qkv = qkv; a.vt = nullptr; a.kbits = kbits; a.out = out;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.Spad = (S + 63) / 64 * 64; a.nw64 = a.Spad / 64;
    a.scale_log2e = 1.4426950408889634f / 11.313708f;
    return fs2::launch_attention_pipe(a, variant, (hipStream_t)stream);
}
#endif
