// Fused multi-head self-attention core for gfx950: softmax(q k^T / sqrt(d) + key-padding) v,
// flash style (scores never reach HBM).  Replaces the attention inside nn.MultiheadAttention as
// nn.TransformerEncoderLayer._sa_block calls it from ConformerEncoderLayer.forward
// (/root/reference/litfass/fastspeech2/model.py:108-116): padded KEYS get -inf, padded queries are
// still computed.
//
// One workgroup = NW waves = NW*32 queries of one (utterance, head); KV advances in tiles of 64
// keys (bf16) / 32 keys (fp32).  K and V tiles are streamed global->LDS by buffer-load DMA (no staging
// registers; a descriptor per utterance, constant per-lane offsets, out-of-range keys read zeros)
// into two single buffers with a 16-byte XOR swizzle applied on the source address: V_j lands
// underneath Q.K^T_j and K_{j+1} underneath P.V_j (two barriers per tile).
//   S^T = K Q^T  via 32x32 MFMA with K as the row operand: each lane then owns ONE query (lane&31)
//                and 16 keys per 32-key block, so row max / row sum are in-register + one lane^32
//                exchange (v_permlane32_swap).  The running max is the accumulators' initial value:
//                p = exp2(acc) with no per-element subtract; the rescale is deferred (threshold).
//   O^T = V^T P^T with V^T as the row operand and the lane's own P registers as the column
//                operand (no cross-lane movement): the MFMA's k-slot <-> key assignment is whatever
//                the S^T register layout gives.  bf16: V stays row-major as it sits in the packed qkv
//                and is read with the hardware transpose read (ds_read_b64_tr_b16); fp32: a V^T scratch
//                with its keys pre-permuted to match (transpose_v kernel), one ds_read_b128 each.
//   K and V fragments are prefetched 6 / 4 MFMAs ahead in a pinned issue order (hipcc otherwise keeps
//   two in flight and the wave runs at LDS latency).
// Key padding arrives as a 64-bit valid mask per 64-key group (any mask shape, not only suffix
// padding); fully padded tiles are skipped.
#include <type_traits>

#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

template <int RB>  // RB = row bytes
__device__ inline int swz_row(int row, int slot) {
    constexpr int NS = RB / 16;
    if constexpr (NS >= 16) return row * RB + ((slot ^ (row & 15)) << 4);
    else if constexpr (NS == 8) return row * RB + ((slot ^ ((row >> 1) & 7)) << 4);
    else if constexpr (NS == 4) return row * RB + ((slot ^ ((row >> 2) & 3)) << 4);
    else return row * RB + ((slot ^ ((row >> 3) & 1)) << 4);
}

// 16-byte global -> LDS DMA: LDS address = (wave-uniform) lds_base + lane*16; the global source
// address is per lane, which is where the XOR swizzle goes (linear destination, permuted source,
// same XOR on the ds_read side — cdna_hip_programming.md §5.4 rule 21).
__device__ inline void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int RB>  // inverse view of swz_row: which logical slot lives at physical slot ps of `row`
__device__ inline int unswz_slot(int row, int ps) {
    constexpr int NS = RB / 16;
    if constexpr (NS >= 16) return ps ^ (row & 15);
    else if constexpr (NS == 8) return ps ^ ((row >> 1) & 7);
    else if constexpr (NS == 4) return ps ^ ((row >> 2) & 3);
    else return ps ^ ((row >> 3) & 1);
}

// Row-major V tile read with ds_read_b64_tr_b16 (bf16 path): a 16-lane group fetches a
// 4 (keys) x 16 (dv) block and every lane receives one dv column = 4 consecutive keys.  The four
// key rows of a block sit 1 row apart (same banks when a row is >= 256 B), so the 16-byte slot is
// XORed with a function of (key & 3) that spreads them over the 256-byte bank window.
template <int RB>
__device__ inline int vswz(int row) {
    constexpr int NS = RB / 16;
    if constexpr (NS >= 16) return (row & 3) << 2;
    else if constexpr (NS == 8) return ((row >> 1) & 1) << 2;
    else return 0;
}
typedef __attribute__((ext_vector_type(4))) short tr_s4;
__device__ inline uint2 tr_read_b64(const unsigned char* lds_addr) {  // lane i of a 16-group -> column i
    const tr_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s4*)lds_addr);
    return *(const uint2*)&v;
}

// NW = waves along the queries (32 each).  (A variant with a second group of waves working on the
// other half of the KV tiles - 8-wave workgroups, one per CU - was measured 7 % slower than two
// independent 4-wave workgroups per CU and retired.)
// Combine a value with its lane ^ 32 partner through v_permlane32_swap (a VALU op, gfx950) instead
// of a ds_bpermute round trip through the LDS: after swapping the upper half of one copy with the
// lower half of another, every lane holds {own, partner} in the two results.
__device__ inline float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// X3 (r04; T = bf16 operands, fp32 output): the parity-grade arithmetic of the fp32x3 / mixed3 modes.  q, k, v arrive as TWO bf16
// tensors each - head and tail of the fp32 values, x = hi + lo up to 2^-17 |x|, written by the in-projection's own store
// (GemmArgs::C_lo) or by split_hi_lo_kernel - and every product is three bf16 MFMAs, lo*hi + hi*lo + hi*hi in the fp32 accumulator
// (small terms first; the dropped lo*lo is 2^-16 of the product), P split the same way in registers.  Softmax, running max, the
// denominator and O stay fp32.  Against the fp32-MFMA form of this kernel (32x32x2, 1/16 of the bf16 rate): 3/16 of the matrix
// time; the decoder launch at C2 went 714 us -> see profiles/HISTORY.md §5.
template <typename T, int D, int NW, bool X3 = false>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 2 && D == 128) ? 2 : 1) void attention_kernel(AttnArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
    constexpr int KVB = sizeof(T) == 2 ? 64 : 32;      // keys per tile
    constexpr int E16 = Num<T>::kPer16B;
    constexpr int KC = Mma32<T>::K_PER_CHUNK;          // k-values per 2x16B chunk
    constexpr int NQC = D / KC;                        // q chunks (16 B per lane each)
    constexpr int KRB = D * (int)sizeof(T);            // K tile row bytes
    constexpr int KNS = KRB / 16;
    constexpr int VRB = 128;                           // Vt tile row bytes (KVB keys)
    constexpr int NKB = KVB / 32;                      // 32-key blocks per tile
    constexpr int ND = D / 32;                         // 32-wide dv blocks
    constexpr int TILE_B = 128 * D;                    // bytes of a K tile == bytes of a Vt tile
    // K / V tile DMAs: 16 KiB / 1 KiB = 16 pieces per tile, issued by the first NDW waves (all of them when the count
    // divides evenly; the 6-wave form lets waves 0-3 issue 4 pieces each)
    constexpr int NDW = (TILE_B / 1024) % NW == 0 ? NW : 4;
    constexpr int NINST = TILE_B / 1024 / NDW;         // 1-KiB DMA instructions per issuing wave per tile
    constexpr float THR = sizeof(T) == 2 ? 6.0f : 0.0f;  // defer-max threshold (log2 units), exact in fp32
    constexpr bool TRV = sizeof(T) == 2;  // bf16: V stays row-major (straight from qkv), hardware transpose read
    static_assert(TILE_B % (1024 * NDW) == 0 && NDW <= NW, "tile must split into whole wave DMAs");

    static_assert(!X3 || TRV, "split arithmetic: bf16 head / tail operands");
    __shared__ __attribute__((aligned(16))) unsigned char sKa[TILE_B];
    __shared__ __attribute__((aligned(16))) unsigned char sVa[TILE_B];
    __shared__ __attribute__((aligned(16))) unsigned char sKl[X3 ? TILE_B : 16];  // the tails' tiles, same layout
    __shared__ __attribute__((aligned(16))) unsigned char sVl[X3 ? TILE_B : 16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // query group (also this wave's share of the tile DMAs)
    const int li = lane & 31, hi = lane >> 5;
    // XCD-aware mapping: workgroup id % 8 is the XCD it lands on (observed dispatch order); all
    // query blocks of one (utterance, head) share its K/V, so keep them on one XCD's L2.
    const int nq = (p.S + NW * 32 - 1) / (NW * 32);
    const int xcd = blockIdx.x & 7, r8 = blockIdx.x >> 3;
    const int bh = (r8 / nq) * 8 + xcd;
    if (bh >= p.B * p.heads) return;
    const int b = bh / p.heads, h = bh % p.heads;
    const int q0 = (r8 % nq) * (NW * 32);
    const int ld = 3 * p.H;
    const T* __restrict__ qkv = (const T*)p.qkv;
    const T* __restrict__ qkvl = (const T*)p.qkv_lo;  // X3 only
    const uint64_t* kbits = p.kbits + (size_t)b * p.nw64;
    const int ntiles = (p.S + KVB - 1) / KVB;

    auto tile_bits = [&](int j) -> unsigned long long {
        unsigned long long w = kbits[(j * KVB) >> 6];
        if (KVB == 32) w = (w >> ((j * KVB) & 32)) & 0xffffffffull;
        return w;
    };
    // K / V tiles come in by buffer-load-to-LDS DMA: one descriptor over THIS utterance's qkv rows
    // (SGPRs), a per-lane byte offset that is constant up to the tile advance, no per-tile address
    // arithmetic beyond one add.  Keys past the end of the utterance fall outside the descriptor's
    // range: the hardware bounds check (on the VGPR offset) returns zeros for them, and they are
    // masked by the valid-key words anyway.
    const unsigned utt_bytes = (unsigned)((size_t)p.S * ld * sizeof(T));
    const __amdgpu_buffer_rsrc_t qrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (size_t)b * p.S * ld), 0, utt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrl =
        __builtin_amdgcn_make_buffer_rsrc((void*)((X3 ? qkvl : qkv) + (size_t)b * p.S * ld), 0, utt_bytes, 0x00020000);
    const unsigned vt_bytes = TRV ? 16u : (unsigned)((size_t)p.B * p.heads * D * p.Spad * sizeof(T));
    const __amdgpu_buffer_rsrc_t vrs =
        __builtin_amdgcn_make_buffer_rsrc(TRV ? (void*)p.qkv : (void*)p.vt, 0, vt_bytes, 0x00020000);
    unsigned kvo[NINST], vvo[NINST];
#pragma unroll
    for (int i = 0; i < NINST; ++i) {
        const int P = (i * NDW + (wave % NDW)) * 64 + lane;
        const int row = P / KNS, ps = P % KNS;
        kvo[i] = (unsigned)((row * ld + p.H + h * D) * (int)sizeof(T) + (unswz_slot<KRB>(row, ps) << 4));
        if constexpr (TRV) {
            vvo[i] = (unsigned)((row * ld + 2 * p.H + h * D) * (int)sizeof(T) + ((ps ^ vswz<KRB>(row)) << 4));
        } else {
            const int vrow = P >> 3, vps = P & 7;
            vvo[i] = (unsigned)(((bh * D + vrow) * p.Spad) * (int)sizeof(T) + (unswz_slot<VRB>(vrow, vps) << 4));
        }
    }
    const unsigned ktile = (unsigned)(KVB * ld * (int)sizeof(T));  // byte advance of the qkv rows per KV tile
    auto dma16 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned voff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    };
    auto issue_k = [&](int j, unsigned char* sK) {
        if (NDW != NW && wave >= NDW) return;
#pragma unroll
        for (int i = 0; i < NINST; ++i) dma16(qrs, sK + (i * NDW + wave) * 1024, kvo[i] + (unsigned)j * ktile);
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < NINST; ++i) dma16(qrl, sKl + (i * NDW + wave) * 1024, kvo[i] + (unsigned)j * ktile);
        }
    };
    auto issue_v = [&](int j, unsigned char* sV) {
        if (NDW != NW && wave >= NDW) return;
#pragma unroll
        for (int i = 0; i < NINST; ++i) {
            if constexpr (TRV) dma16(qrs, sV + (i * NDW + wave) * 1024, vvo[i] + (unsigned)j * ktile);
            else dma16(vrs, sV + (i * NDW + wave) * 1024, vvo[i] + (unsigned)(j * KVB * (int)sizeof(T)));
        }
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < NINST; ++i) dma16(qrl, sVl + (i * NDW + wave) * 1024, vvo[i] + (unsigned)j * ktile);
        }
    };

    // ---- Q fragments (column operand of S^T = K Q^T) ----
    uint4 qf[NQC], qfl[X3 ? NQC : 1];
    {
        int qrow = q0 + wave * 32 + li;
        if (qrow >= p.S) qrow = p.S - 1;
        const T* src = qkv + (size_t)(b * p.S + qrow) * ld + h * D + hi * E16;
#pragma unroll
        for (int c = 0; c < NQC; ++c) {  // q * log2(e)/sqrt(d), once: scores come out of the MFMA in exp2 units
            float f[Vec16<T>::N];
            Vec16<T>::unpack(*(const uint4*)(src + c * KC), f);
            if constexpr (X3) {  // the fp32 value back from its two halves, scaled in fp32, split again
                float g[Vec16<T>::N], r[Vec16<T>::N];
                Vec16<T>::unpack(*(const uint4*)(qkvl + (size_t)(b * p.S + qrow) * ld + h * D + hi * E16 + c * KC), g);
#pragma unroll
                for (int e = 0; e < Vec16<T>::N; ++e) f[e] = (f[e] + g[e]) * p.scale_log2e;
                qf[c] = Vec16<T>::pack(f);
                Vec16<T>::unpack(qf[c], g);
#pragma unroll
                for (int e = 0; e < Vec16<T>::N; ++e) r[e] = f[e] - g[e];
                qfl[c] = Vec16<T>::pack(r);
            } else {
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) f[e] *= p.scale_log2e;
            qf[c] = Vec16<T>::pack(f);
            }
        }
    }

    f32x16_t oacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;  // running max (scaled, log2 units) and denominator

    // This half's tile range; all halves run the same number of (two-barrier) iterations.
    const int nhalf = ntiles, jbeg = 0, jend = ntiles;
    auto run = [&](unsigned char* sK0, unsigned char* sV0) {
        if (jbeg < jend && tile_bits(jbeg) != 0ull) issue_k(jbeg, sK0);
        for (int it = 0; it < nhalf; ++it) {
            const int j = jbeg + it;
            const unsigned long long bits = j < jend ? tile_bits(j) : 0ull;
            const bool valid = bits != 0ull;  // fully padded tiles cost two barriers, nothing else
            unsigned char* const sK = sK0;
            unsigned char* const sV = sV0;
            dma_drain();       // this wave's share of K_j has landed (explicit: never left to hipcc's
            __syncthreads();   // placement); after the barrier everyone's has, and every wave is
                               // done with P.V of the previous tile -> sV is free
            if (valid) issue_v(j, sV);  // V_j streams in underneath Q.K^T
            uint4 pf[4], pfl[X3 ? 4 : 1];
            if (valid) {
            // the running max rides in the accumulator's initial value, so the common path is
            // p = exp2(acc) with no per-element subtract
            const float m_eff = (m_run == -INFINITY) ? 0.f : m_run;
            const float m_init = -m_eff;
            // ---- S^T = K Q^T ----
            f32x16_t sacc[NKB];
    #pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = m_init;  // accumulate (s - m) directly
            // chunk-outer / key-block-inner: consecutive MFMAs hit DIFFERENT accumulators, so the
            // accumulate latency of one chain hides under the other
            {   // rolling prefetch: PD K fragments are in flight ahead of the MFMA that consumes them, so a
                // wave's Q.K^T is paced by the matrix pipe, not by an LDS round trip per MFMA pair
                constexpr int NF = NQC * NKB, PD = X3 ? (NF < 3 ? NF : 3) : (NF < 6 ? NF : 6);
                uint4 kq[PD], kql[X3 ? PD : 1];
                auto koff = [&](int i) { return swz_row<KRB>((i % NKB) * 32 + li, (i / NKB) * 2 + hi); };
                auto kaddr = [&](int i) { return sK + koff(i); };
    #pragma unroll
                for (int i = 0; i < PD; ++i) {
                    kq[i] = *(const uint4*)kaddr(i);
                    if constexpr (X3) kql[i] = *(const uint4*)(sKl + koff(i));
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the issue order: hipcc otherwise re-serialises to 2 in flight
    #pragma unroll
                for (int i = 0; i < NF; ++i) {
                    if constexpr (X3) {  // small terms first
                        Mma32<T>::step(kql[i % PD], qf[i / NKB], sacc[i % NKB]);
                        Mma32<T>::step(kq[i % PD], qfl[i / NKB], sacc[i % NKB]);
                    }
                    Mma32<T>::step(kq[i % PD], qf[i / NKB], sacc[i % NKB]);
                    if (i + PD < NF) {
                        kq[i % PD] = *(const uint4*)kaddr(i + PD);
                        if constexpr (X3) kql[i % PD] = *(const uint4*)(sKl + koff(i + PD));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- key-padding mask (only tiles that contain a padded key pay for it) ----
            const unsigned long long full = KVB == 64 ? ~0ull : 0xffffffffull;
            if (bits != full) {
    #pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (!((bits >> ko) & 1ull)) sacc[kb][r] = -INFINITY;
                    }
            }
            // ---- online softmax, base 2: acc holds s - m_eff ----
            float mx = sacc[0][0];
    #pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            mx = half_max(mx);  // growth of the row max relative to m_eff (the other 16 keys: lane ^ 32)
            float delta = 0.f;
            if (__any(m_run == -INFINITY || mx > THR)) {  // first tile of a row, or its max grew past the threshold
                const float m_new = fmaxf(m_run, m_eff + mx);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);  // m_run=-inf -> 0
                delta = (m_new == -INFINITY) ? 0.f : m_new - m_eff;
                l_run *= alpha;
    #pragma unroll
                for (int i = 0; i < ND; ++i)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
                m_run = m_new;
    #pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
            }
            float rs = 0.f;
    #pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sacc[kb][r]);
                    sacc[kb][r] = e;
                    rs += e;
                }
            rs = half_sum(rs);
            l_run += rs;

            // ---- attention-weight dropout (training): after the row sum, before P.V (the softmax normalisation is of the
            // undropped weights); mask = f(seed, key, ((b*heads + h)*S + query)*S + key index) ----
            if (p.drop_p > 0.f) {
                const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
                const float dsc = 1.f / (1.f - p.drop_p);
                const uint64_t rowbase = ((uint64_t)(b * p.heads + h) * p.S + (q0 + wave * 32 + li)) * p.S + (uint64_t)j * KVB;
    #pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        sacc[kb][r] = dropout_bits(p.drop_seed, p.drop_key, rowbase + ko) >= thr ? sacc[kb][r] * dsc : 0.f;
                    }
            }
            // ---- P fragments (column operand), straight from the lane's own registers ----
    #pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float f[Vec16<T>::N];
    #pragma unroll
                for (int e = 0; e < Vec16<T>::N; ++e) {
                    const int flat = ch * Vec16<T>::N + e;
                    f[e] = sacc[flat >> 4][flat & 15];
                }
                pf[ch] = Vec16<T>::pack(f);
                if constexpr (X3) {  // p = head + tail
                    float g[Vec16<T>::N];
                    Vec16<T>::unpack(pf[ch], g);
    #pragma unroll
                    for (int e = 0; e < Vec16<T>::N; ++e) g[e] = f[e] - g[e];
                    pfl[ch] = Vec16<T>::pack(g);
                }
            }
            }
            dma_drain();
            __syncthreads();   // V_j landed; every wave is done reading sK
            if (j + 1 < jend && tile_bits(j + 1) != 0ull) issue_k(j + 1, sK);  // next K under P.V
            if (valid) {
            // ---- O^T += V^T P^T ----
            if constexpr (TRV) {
                // lane (dv = lane&31, hi): elements 0..3 <- keys ch*16 + 4hi + 0..3, 4..7 <- +8: the same
                // k-slot <-> key map the P registers carry
                const int i16 = lane & 15, g1 = (lane >> 4) & 1;
                const int rsub = i16 >> 2;                      // key row inside the 4-row block (== key & 3)
                const int rowb = (4 * hi + rsub) * KRB + (i16 & 1) * 8;
                int vcol[ND];
    #pragma unroll
                for (int nd = 0; nd < ND; ++nd)
                    vcol[nd] = rowb + (((nd * 4 + g1 * 2 + ((i16 & 3) >> 1)) ^ vswz<KRB>(rsub)) << 4);
                // key-chunk outer / dv-block inner: consecutive MFMAs accumulate into different oacc[nd]
                {   // same rolling prefetch for the transposed V fragments (two 8-byte reads each)
                    constexpr int NF = 4 * ND, PD = X3 ? 2 : 4;
                    uint4 vq[PD], vql[X3 ? PD : 1];
                    auto vload = [&](const unsigned char* base, int i) {
                        const unsigned char* vb = base + vcol[i % ND] + (i / ND) * 16 * KRB;
                        const uint2 lo = tr_read_b64(vb);
                        const uint2 hi2 = tr_read_b64(vb + 8 * KRB);
                        return make_uint4(lo.x, lo.y, hi2.x, hi2.y);
                    };
    #pragma unroll
                    for (int i = 0; i < PD; ++i) {
                        vq[i] = vload(sV, i);
                        if constexpr (X3) vql[i] = vload(sVl, i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        if constexpr (X3) {
                            Mma32<T>::step(vql[i % PD], pf[i / ND], oacc[i % ND]);
                            Mma32<T>::step(vq[i % PD], pfl[i / ND], oacc[i % ND]);
                        }
                        Mma32<T>::step(vq[i % PD], pf[i / ND], oacc[i % ND]);
                        if (i + PD < NF) {
                            vq[i % PD] = vload(sV, i + PD);
                            if constexpr (X3) vql[i % PD] = vload(sVl, i + PD);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
    #pragma unroll
                for (int ch = 0; ch < 4; ++ch)
    #pragma unroll
                    for (int nd = 0; nd < ND; ++nd) {
                        const uint4 vf = *(const uint4*)(sV + swz_row<VRB>(nd * 32 + li, ch * 2 + hi));
                        Mma32<T>::step(vf, pf[ch], oacc[nd]);
                    }
            }
            }
        }
    };
    run(sKa, sVa);

    // ---- normalise and store: lane owns query li, dv = nd*32 + (r&3) + 8*(r>>2) + 4*hi ----
    const int qrow = q0 + wave * 32 + li;
    if (qrow < p.S) {
        if (p.lse2 && hi == 0) p.lse2[(size_t)(b * p.heads + h) * p.S + qrow] = m_run + __builtin_amdgcn_logf(l_run);  // v_log_f32 = log2
        const float inv = 1.f / l_run;  // all keys padded -> NaN, as the reference's softmax gives
        using OT = typename std::conditional<X3, float, T>::type;  // the split form hands back fp32 rows
        OT* dst = (OT*)p.out + (size_t)(b * p.S + qrow) * p.H + h * D;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dv = nd * 32 + 8 * g + 4 * hi;
                const float v0 = oacc[nd][4 * g + 0] * inv, v1 = oacc[nd][4 * g + 1] * inv;
                const float v2 = oacc[nd][4 * g + 2] * inv, v3 = oacc[nd][4 * g + 3] * inv;
                if constexpr (sizeof(OT) == 4) {
                    *(float4*)(dst + dv) = make_float4(v0, v1, v2, v3);
                } else {
                    *(uint2*)(dst + dv) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
            }
    }
#else
    (void)p;
#endif
}

// =====================================================================================================================================
// r06: the encoder's self-attention + out-projection + residual + LayerNorm as ONE launch (bf16, H = 256, two heads of 128) - two thirds
// of the block BASELINE.json's north star names (nn.MultiheadAttention inside ConformerEncoderLayer.forward, model.py:108-116; the
// in-projection stays a GEMM launch in front).  A 4-wave workgroup owns 64 queries of one utterance: waves 0-1 run head 0, waves 2-3 head 1 -
// each pair exactly attention_kernel<bf16, 128, 2>'s loop (same per-row instruction sequence: O is the same bits) on its own K / V tiles, in
// lock step (same utterance, same key mask, same barriers).  The normalised O rows (64 x 256 bf16) go into LDS where the K tiles were, in
// the single-launch predictor's slab layout, and become the A operand of the out-projection: each wave owns all 64 rows x 64 output
// channels, the 256 x 256 weights stream from L2 straight into MFMA fragments (packed at fs2_finalize in predictor_fused.hip's fragment
// order, launch_pack_predictor_weights(taps = 1)), the bias is the
// accumulators' initial value, the residual rows are requested before the K loop, LayerNorm as in the predictor's epilogue (two-pass
// statistics, lane-group sums by permlane swaps, one LDS exchange between the four waves per pass).  Replaces a 256-workgroup attention
// launch + a 256-workgroup GEMM + LayerNorm launch (10.7 + 14.8 us at C2: 5.2 k CU-us) by 128 workgroups.
struct AttnOutSmem {
    static constexpr int TILE_B = 128 * 128;  // a 64-key x 128-dim bf16 K (or row-major V) tile
};
__global__ __launch_bounds__(256, 2) void attn_out_ln_kernel(AttnOutArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16;
    constexpr int D = 128, HH = 256, KVB = 64, E16 = 8, KC = Mma32<T>::K_PER_CHUNK, NQC = D / KC, KRB = D * 2, KNS = KRB / 16;
    constexpr int NKB = KVB / 32, ND = D / 32, TILE_B = AttnOutSmem::TILE_B, NINST = TILE_B / 1024 / 2;
    constexpr float THR = 6.0f;
    // (measured and dropped, r06: K and V tiles double-buffered - K_{j+1} / V_{j+1} requested a whole tile ahead, one barrier per tile, four
    //  LDS objects with static names, 133 KB - 22.3 us against 20.2 for this two-barrier form in tools/bench_ops.py encmha: the loop is not
    //  bound by the tiles' round trips, as r02's one-barrier ring and r03's resident-K/V form of attention_kernel had already measured)
    __shared__ __attribute__((aligned(16))) unsigned char sK[2 * TILE_B];  // [head]; afterwards: the O slab (64 rows x 512 B)
    __shared__ __attribute__((aligned(16))) unsigned char sV[2 * TILE_B];
    __shared__ float red[2][4 * 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 1, wq = wave & 1;  // head; query group (32 queries) inside the head's pair of waves
    const int li = lane & 31, hi = lane >> 5;
    const int nq = (p.S + 63) / 64;
    const int xcd = blockIdx.x & 7, r8 = blockIdx.x >> 3;  // all query blocks of an utterance on one XCD's L2
    const int b = (r8 / nq) * 8 + xcd;
    if (b >= p.B) return;
    const int q0 = (r8 % nq) * 64;
    const int ld = 3 * HH;
    const T* __restrict__ qkv = (const T*)p.qkv;
    const uint64_t* kbits = p.kbits + (size_t)b * p.nw64;
    const int ntiles = (p.S + KVB - 1) / KVB;

    // ---- out-projection weight stream: this wave's 4 fragments per 32-k step, 4-stage ring, first three stages requested now ----
    constexpr int NFR = 4, MI16 = 4, PKB = HH / 32, STEP_U4 = (HH / 32) * 2 * 64, RING = PKB;  // the whole 8-step stream in flight at once (128 registers: O's are dead by then)
    const uint4* __restrict__ wbase = (const uint4*)p.wpk + wave * NFR * 64 + lane;
    uint4 bw[RING][NFR];
    auto loadB = [&](uint4 (&bb)[NFR], int g) {
        g = g < PKB ? g : PKB - 1;
#pragma unroll
        for (int ni = 0; ni < NFR; ++ni) bb[ni] = wbase[(size_t)g * STEP_U4 + ni * 64];
    };

    const unsigned utt_bytes = (unsigned)((size_t)p.S * ld * sizeof(T));
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (size_t)b * p.S * ld), 0, utt_bytes, 0x00020000);
    unsigned kvo[NINST], vvo[NINST];
#pragma unroll
    for (int i = 0; i < NINST; ++i) {
        const int P = (i * 2 + wq) * 64 + lane;
        const int row = P / KNS, ps = P % KNS;
        kvo[i] = (unsigned)((row * ld + HH + h * D) * 2 + (unswz_slot<KRB>(row, ps) << 4));
        vvo[i] = (unsigned)((row * ld + 2 * HH + h * D) * 2 + ((ps ^ vswz<KRB>(row)) << 4));
    }
    const unsigned ktile = (unsigned)(KVB * ld * 2);
    auto dma16 = [&](unsigned char* dst, unsigned voff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    };
    auto issue_k = [&](int j, unsigned char* buf) {
        unsigned char* const d = buf + h * TILE_B;
#pragma unroll
        for (int i = 0; i < NINST; ++i) dma16(d + (i * 2 + wq) * 1024, kvo[i] + (unsigned)j * ktile);
    };
    auto issue_v = [&](int j, unsigned char* buf) {
        unsigned char* const d = buf + h * TILE_B;
#pragma unroll
        for (int i = 0; i < NINST; ++i) dma16(d + (i * 2 + wq) * 1024, vvo[i] + (unsigned)j * ktile);
    };
    auto tile_bits = [&](int j) -> unsigned long long { return kbits[(j * KVB) >> 6]; };

    // ---- Q fragments, scaled by log2(e) / sqrt(d) ----
    uint4 qf[NQC];
    {
        int qrow = q0 + wq * 32 + li;
        if (qrow >= p.S) qrow = p.S - 1;
        const T* src = qkv + (size_t)(b * p.S + qrow) * ld + h * D + hi * E16;
#pragma unroll
        for (int c = 0; c < NQC; ++c) {
            float f[8];
            Vec16<T>::unpack(*(const uint4*)(src + c * KC), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2e;
            qf[c] = Vec16<T>::pack(f);
        }
    }
    f32x16_t oacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- the attention loop: attention_kernel<bf16, 128, 2>'s, per head pair ----
    const unsigned char* const sKh = sK + h * TILE_B;
    const unsigned char* const sVh = sV + h * TILE_B;
    if (ntiles > 0 && tile_bits(0) != 0ull) issue_k(0, sK);
    for (int j = 0; j < ntiles; ++j) {
        const unsigned long long bits = tile_bits(j);
        const bool valid = bits != 0ull;
        dma_drain();
        __syncthreads();  // K_j has landed; everyone is past P.V of tile j - 1: sV is free
        if (valid) issue_v(j, sV);  // V_j streams in underneath Q.K^T
        uint4 pf[4];
        if (valid) {
            const float m_eff = (m_run == -INFINITY) ? 0.f : m_run;
            const float m_init = -m_eff;
            f32x16_t sacc[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = m_init;
            {
                constexpr int NF = NQC * NKB, PD = NF < 6 ? NF : 6;
                uint4 kq[PD];
                auto kaddr = [&](int i) { return sKh + swz_row<KRB>((i % NKB) * 32 + li, (i / NKB) * 2 + hi); };
#pragma unroll
                for (int i = 0; i < PD; ++i) kq[i] = *(const uint4*)kaddr(i);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    Mma32<T>::step(kq[i % PD], qf[i / NKB], sacc[i % NKB]);
                    if (i + PD < NF) kq[i % PD] = *(const uint4*)kaddr(i + PD);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (bits != ~0ull) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (!((bits >> ko) & 1ull)) sacc[kb][r] = -INFINITY;
                    }
            }
            float mx = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            mx = half_max(mx);
            if (__any(m_run == -INFINITY || mx > THR)) {
                const float m_new = fmaxf(m_run, m_eff + mx);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
                const float delta = (m_new == -INFINITY) ? 0.f : m_new - m_eff;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
                m_run = m_new;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
            }
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sacc[kb][r]);
                    sacc[kb][r] = e;
                    rs += e;
                }
            rs = half_sum(rs);
            l_run += rs;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int flat = ch * 8 + e;
                    f[e] = sacc[flat >> 4][flat & 15];
                }
                pf[ch] = Vec16<T>::pack(f);
            }
        }
        dma_drain();
        __syncthreads();  // V_j has landed; every wave is done reading sK
        if (j + 1 < ntiles && tile_bits(j + 1) != 0ull) issue_k(j + 1, sK);  // next K under P.V
        if (valid) {
            const int i16 = lane & 15, g1 = (lane >> 4) & 1;
            const int rsub = i16 >> 2;
            const int rowb = (4 * hi + rsub) * KRB + (i16 & 1) * 8;
            int vcol[ND];
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) vcol[nd] = rowb + (((nd * 4 + g1 * 2 + ((i16 & 3) >> 1)) ^ vswz<KRB>(rsub)) << 4);
            constexpr int NF = 4 * ND, PD = 4;
            uint4 vq[PD];
            auto vload = [&](int i) {
                const unsigned char* vb = sVh + vcol[i % ND] + (i / ND) * 16 * KRB;
                const uint2 lo = tr_read_b64(vb);
                const uint2 hi2 = tr_read_b64(vb + 8 * KRB);
                return make_uint4(lo.x, lo.y, hi2.x, hi2.y);
            };
#pragma unroll
            for (int i = 0; i < PD; ++i) vq[i] = vload(i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                Mma32<T>::step(vq[i % PD], pf[i / ND], oacc[i % ND]);
                if (i + PD < NF) vq[i % PD] = vload(i + PD);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < RING; ++g) loadB(bw[g], g);  // (requested here, not at the top: registers held across the attention loop spill)
    // ---- O rows -> the slab (where the K tiles were: nobody reads sK after the loop's last barrier) ----
    // lane owns query row wq*32 + li and channels h*128 + nd*32 + 8g + 4hi + 0..3: half a 16-byte slot
    const SlabSwizzle sw(HH * 2 / 16);
    unsigned char* const slab = sK;
    {
        const float inv = 1.f / l_run;  // all keys padded -> NaN, as the reference's softmax gives
        const int row = wq * 32 + li;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int L = h * 16 + nd * 4 + g;
                const uint2 o2 = make_uint2(pack_bf16x2(oacc[nd][4 * g + 0] * inv, oacc[nd][4 * g + 1] * inv),
                                            pack_bf16x2(oacc[nd][4 * g + 2] * inv, oacc[nd][4 * g + 3] * inv));
                *(uint2*)(slab + row * (HH * 2) + (sw.slot(L, row) << 4) + hi * 8) = o2;
            }
    }
    // ---- residual rows (requested before the K loop) and bias ----
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = wave * (NFR * 16) + fg * 8;  // fragment pair j: this lane's channels n0 + 32 j .. + 7
    uint4 resv[MI16][2];
    {
        const T* rs_ = (const T*)p.res + (size_t)b * p.S * HH;
#pragma unroll
        for (int m = 0; m < MI16; ++m) {
            int t = q0 + m * 16 + fr;
            t = t < p.S ? t : p.S - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) resv[m][j] = *(const uint4*)(rs_ + (size_t)t * HH + n0 + 32 * j);
        }
    }
    f32x4_t acc[NFR][MI16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 b0 = *(const float4*)(p.bias + n0 + 32 * j), b1 = *(const float4*)(p.bias + n0 + 32 * j + 4);
#pragma unroll
        for (int m = 0; m < MI16; ++m) {
            acc[2 * j][m] = (f32x4_t){b0.x, b0.y, b0.z, b0.w};
            acc[2 * j + 1][m] = (f32x4_t){b1.x, b1.y, b1.z, b1.w};
        }
    }
    __syncthreads();  // the slab holds O
    // ---- out-projection: 8 steps of 32 k ----
    {
        const unsigned char* arow_p = slab + fr * (HH * 2);
        const int acx = (((fg & 1) << 3) | ((fg >> 1) ^ (fr & 7))) << 4;  // SlabSwizzle::slot(fg, fr); + kb below
#pragma unroll
        for (int kb = 0; kb < PKB; ++kb) {
            uint4 fx[MI16];
#pragma unroll
            for (int mi = 0; mi < MI16; ++mi) fx[mi] = *(const uint4*)(arow_p + mi * 16 * (HH * 2) + (acx ^ ((((kb & 3) << 1) | ((kb >> 2) << 4)) << 4)));
#pragma unroll
            for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI16; ++mi) Mma16<T>::step(bw[kb % RING][ni], fx[mi], acc[ni][mi]);
        }
    }
    // ---- + residual, LayerNorm (the single-launch predictor's epilogue without the ReLU), store ----
    const float invn = 1.0f / (float)HH;
    float mean[MI16], rstd[MI16];
#pragma unroll
    for (int m = 0; m < MI16; ++m) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float rf[8];
            Vec16<T>::unpack(resv[m][j], rf);
            f32x4_t& a0 = acc[2 * j][m];
            f32x4_t& a1 = acc[2 * j + 1][m];
            a0 = (f32x4_t){a0[0] + rf[0], a0[1] + rf[1], a0[2] + rf[2], a0[3] + rf[3]};
            a1 = (f32x4_t){a1[0] + rf[4], a1[1] + rf[5], a1[2] + rf[6], a1[3] + rf[7]};
            s += (a0[0] + a0[1]) + (a0[2] + a0[3]) + (a1[0] + a1[1]) + (a1[2] + a1[3]);
        }
        s = group4_sum(s);
        if (fg == 0) red[0][wave * 64 + m * 16 + fr] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MI16; ++m) {
        const int row = m * 16 + fr;
        const float t4 = (red[0][row] + red[0][64 + row]) + (red[0][128 + row] + red[0][192 + row]);
        mean[m] = t4 * invn;
        float q = 0.f;
#pragma unroll
        for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[ni][m][r] - mean[m];
                acc[ni][m][r] = d;
                q = __builtin_fmaf(d, d, q);
            }
        q = group4_sum(q);
        if (fg == 0) red[1][wave * 64 + row] = q;
    }
    __syncthreads();
    T* out = (T*)p.out + (size_t)b * p.S * HH;
#pragma unroll
    for (int m = 0; m < MI16; ++m) {
        const int row = m * 16 + fr, t = q0 + row;
        const float t4 = (red[1][row] + red[1][64 + row]) + (red[1][128 + row] + red[1][192 + row]);
        rstd[m] = 1.0f / sqrtf(t4 * invn + p.eps);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * j;
            const float4 g0 = *(const float4*)(p.ln_g + n), g1 = *(const float4*)(p.ln_g + n + 4);
            const float4 e0 = *(const float4*)(p.ln_b + n), e1 = *(const float4*)(p.ln_b + n + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            float y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) y[r] = __builtin_fmaf(acc[2 * j + (r >> 2)][m][r & 3] * rstd[m], gg[r], ee[r]);
            if (t < p.S) *(uint4*)(out + (size_t)t * HH + n) = Vec16<T>::pack(y);
        }
    }
#else
    (void)p;
#endif
}

bool attn_out_ln_supported(int dtype, int H, int heads, int S) {
    return dtype == FS2_BF16 && H == 256 && heads == 2 && S >= 1 && (size_t)S * 3 * H * 2 < 0xFFFFF000ull;
}
int launch_attn_out_ln(const AttnOutArgs& a, hipStream_t stream) {
    if (!attn_out_ln_supported(FS2_BF16, a.H, a.heads, a.S)) return FS2_ERR_SHAPE;
    if (!a.qkv || !a.kbits || !a.wpk || !a.bias || !a.res || !a.ln_g || !a.ln_b || !a.out || a.out == a.qkv) return FS2_ERR_ARG;
    if (((uintptr_t)a.qkv | (uintptr_t)a.wpk | (uintptr_t)a.bias | (uintptr_t)a.res | (uintptr_t)a.ln_g | (uintptr_t)a.ln_b | (uintptr_t)a.out) & 15) return FS2_ERR_ARG;
    if (a.B <= 0) return FS2_OK;
    const int nq = (a.S + 63) / 64, B8 = (a.B + 7) / 8 * 8;
    hipLaunchKernelGGL(attn_out_ln_kernel, dim3((unsigned)(B8 * nq)), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

// ---- V^T staging: qkv's V columns -> Vt[(b*heads+h)][dv][Spad], zero padded to Spad; for bf16
// the keys inside each 16-group are permuted to the order the S^T register layout consumes:
//   pos(o) = ((o>>2)&1)*8 + (o&3) + 4*(o>>3).
template <typename T, int D>
__global__ __launch_bounds__(256) void transpose_v_kernel(AttnArgs p) {
    __shared__ float tile[64][D + 1];
    const int bh = blockIdx.y, b = bh / p.heads, h = bh % p.heads;
    const int k0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const T* __restrict__ qkv = (const T*)p.qkv;
    const int ld = 3 * p.H;
    for (int i = tid; i < 64 * D; i += 256) {
        const int key = i / D, dv = i % D;
        float v = 0.f;
        if (k0 + key < p.S) v = Num<T>::to_f32(qkv[(size_t)(b * p.S + k0 + key) * ld + 2 * p.H + h * D + dv]);
        tile[key][dv] = v;
    }
    __syncthreads();
    T* __restrict__ vt = (T*)p.vt + (size_t)bh * D * p.Spad + k0;
    for (int i = tid; i < 64 * D; i += 256) {
        const int dv = i / 64, pos = i % 64;
        int key = pos;
        if (sizeof(T) == 2) {  // inverse of pos(o): within the 16-group, pos = hi*8 + j
            const int g16 = pos & ~15, q = pos & 15, hh = q >> 3, jj = q & 7;
            key = g16 + (jj & 3) + 8 * (jj >> 2) + 4 * hh;
        }
        vt[(size_t)dv * p.Spad + pos] = Num<T>::from_f32(tile[key][dv]);
    }
}

// (r03: a resident-K/V form for <= 256 keys - all of an (utterance, head)'s K and V requested at the top of the kernel into 128 KiB of
// LDS, two barriers in all - was built, bit-identical, and measured no faster: C2 encoder 13.1 us vs 13.6 streaming, C3 encoder 28.1
// vs 19.3; the 128 KiB a workgroup pulls before its first MFMA arrive at the CU's ingest rate whatever the request pattern, and two
// co-resident streaming workgroups hide each other's round trips.  Removed in r05; profiles/HISTORY.md §4 keeps the numbers.)
template <typename T, int D>
static int launch_tv(const AttnArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL((transpose_v_kernel<T, D>), dim3(a.Spad / 64, a.B * a.heads), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

template <typename T, int D>
static int launch_td(const AttnArgs& a, hipStream_t stream) {
    const int BH = a.B * a.heads, BH8 = (BH + 7) / 8 * 8;
    // grid = ceil(BH/8)*8 * nq, decoded XCD-aware in the kernel.  Small sequences: 2-wave
    // workgroups so the grid still covers the 256 CUs.
    const long blocks4 = (long)((a.S + 127) / 128) * BH;
    // (192-query / 6-wave workgroups - 512 of them for the C2 decoder, two per CU, a third less K/V streamed - need three
    // waves per SIMD, i.e. <= 168 VGPRs; this kernel holds 234 (O^T 64, S^T 32, Q 32, K/V/P fragments 56 ...) and with the cap
    // spills 84 of them: 195 us against 101 us for the 128-query form.  Measured r02, removed.)
    if (blocks4 >= 512) {
        hipLaunchKernelGGL((attention_kernel<T, D, 4>), dim3(((a.S + 127) / 128) * BH8), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((attention_kernel<T, D, 2>), dim3(((a.S + 63) / 64) * BH8), dim3(128), 0, stream, a);
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

int launch_transpose_v(const AttnArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.S <= 0) return FS2_OK;
    if (dtype == FS2_BF16) return FS2_OK;  // bf16 attention reads V row-major with the hardware transpose read
    if (a.Spad % 64 || a.Spad < a.S || a.H % a.heads) return FS2_ERR_SHAPE;
    const int d = a.H / a.heads;
#define FS2_TV_CASE(DD)                                                          \
    if (d == DD) return dtype == FS2_BF16 ? launch_tv<bf16, DD>(a, stream) : launch_tv<float, DD>(a, stream);
    FS2_TV_CASE(32)
    FS2_TV_CASE(64)
    FS2_TV_CASE(128)
#undef FS2_TV_CASE
    return FS2_ERR_SHAPE;
}

// fp32 (B*S, 3H) -> bf16 heads and tails, x = hi + lo (the X3 attention's operands when the in-projection did not write them itself)
__global__ __launch_bounds__(256) void split_hi_lo_kernel(const float* __restrict__ x, bf16* __restrict__ hi, bf16* __restrict__ lo, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 c0 = *(const uint4*)(x + i * 8), c1 = *(const uint4*)(x + i * 8 + 4);
    uint4 h, l;
    split_bf16x3(c0, c1, h, l);
    *(uint4*)(hi + i * 8) = h;
    *(uint4*)(lo + i * 8) = l;
}
int launch_split_hi_lo(const float* x, void* hi, void* lo, size_t n, hipStream_t stream) {
    if (n % 8) return FS2_ERR_SHAPE;
    if (!n) return FS2_OK;
    hipLaunchKernelGGL(split_hi_lo_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, x, (bf16*)hi, (bf16*)lo, n / 8);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

template <int D>
static int launch_x3(const AttnArgs& a, hipStream_t stream) {
    const int BH = a.B * a.heads, BH8 = (BH + 7) / 8 * 8;
    const long blocks4 = (long)((a.S + 127) / 128) * BH;
    if (blocks4 >= 512) hipLaunchKernelGGL((attention_kernel<bf16, D, 4, true>), dim3(((a.S + 127) / 128) * BH8), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attention_kernel<bf16, D, 2, true>), dim3(((a.S + 63) / 64) * BH8), dim3(128), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}


int launch_attention(const AttnArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.S <= 0) return FS2_OK;
    if (a.Spad % 64 || a.Spad < a.S || a.H % a.heads) return FS2_ERR_SHAPE;
    if (a.qkv_lo) {  // split arithmetic: bf16 head / tail operands, fp32 rows out (dtype names the STORAGE mode: fp32)
        const int d3 = a.H / a.heads;
        if (dtype != FS2_F32) return FS2_ERR_ARG;
        if (d3 == 32) return launch_x3<32>(a, stream);
        if (d3 == 64) return launch_x3<64>(a, stream);
        if (d3 == 128) return launch_x3<128>(a, stream);
        return FS2_ERR_SHAPE;
    }
    const int g_attn_pipe = tuning_of(a.tune).attn_pipe;
    if (g_attn_pipe && attention_pipe_supported(a, dtype)) {
        // WHICH kernel family computes an utterance may depend on the utterance only (its length, the head count), never on the
        // batch around it: a shard run alone must be bit-equal to its rows of the whole batch (the data-parallel invariant,
        // tests/test_gpu_configs.py).  The two pipelined variants are bit-identical per row (same per-row instruction sequence),
        // so the choice between THEM may follow the launch size: one workgroup per CU with 64 queries per wave from about two
        // 128-query units per CU (measured, r03: C2 / C3 / C5 decoder 90 / 296 / 90 us against 98 / 315 / 93 for the kernel
        // below), two 4 x 32-query workgroups per CU under that.  Short sequences (the encoder's 256 phonemes) stay below.
        const long long per_utt = (long long)a.heads * ((a.S + 127) / 128), units = per_utt * a.B;
        if (g_attn_pipe == 1 || g_attn_pipe == 2) return launch_attention_pipe(a, g_attn_pipe, stream);
        if (g_attn_pipe == 4) return launch_attention_pipe(a, 3, stream);
        // 96 queries per wave (384-query items) from three units per workgroup on: at T = 1536 (12 units per head) a workgroup's
        // run is whole items (one per CU at C2 / C5, three at C3); other lengths finish a head with a 256- or 128-query item and
        // measured equal or faster than the 64-query-per-wave kernel (r03: T = 1400 66 vs 72 us, 1000 / 1664 / 2000 equal)
        if (per_utt >= 16) return launch_attention_pipe(a, units >= 768 ? 3 : (units >= 512 ? 2 : 1), stream);
    }
    const int d = a.H / a.heads;
#define FS2_ATTN_CASE(DD)                                                        \
    if (d == DD) return dtype == FS2_BF16 ? launch_td<bf16, DD>(a, stream) : launch_td<float, DD>(a, stream);
    FS2_ATTN_CASE(32)
    FS2_ATTN_CASE(64)
    FS2_ATTN_CASE(128)
#undef FS2_ATTN_CASE
    return FS2_ERR_SHAPE;
}

}  // namespace fs2
