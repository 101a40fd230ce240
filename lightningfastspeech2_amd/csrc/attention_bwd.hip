// Recomputing (flash) backward of the self-attention core for the training step: dQ, dK, dV from dO, the
// forward's packed qkv and its per-query log-sum-exp - the (B, heads, S, S) probabilities and their gradients never reach
// HBM.  The reference gets this from autograd of nn.MultiheadAttention (model.py:108-116 -> F.multi_head_attention_forward:
// softmax(q k^T / sqrt(d) + key padding) -> dropout -> @ v).  bf16 operands, fp32 accumulation, head dim 128.
//
//   P  = exp2(s * log2(e)/sqrt(d) - lse2)           recomputed per 16 x 16 block (lse2 from the forward)
//   dP = dO V^T ;  dS = P o (dropout(dP) - delta) / sqrt(d) ,  delta = sum_d dO O  (fs2_op_attn_delta)
//   dV = dropout(P)^T dO ;  dK = dS^T Q ;  dQ = dS K
//
// Two launches so that no gradient needs atomics (bit-equal reruns):
//   attn_bwd_dkdv: a workgroup owns 64 keys (a wave 16 of them: its K / V fragments live in registers) and walks the queries in
//                  blocks of 64 (Q and dO tiles in LDS);
//   attn_bwd_dq:   a workgroup owns 64 queries (a wave 16: Q / dO fragments, lse2 and delta in registers) and walks the keys.
// Both form S / dP blocks with 16x16x32 MFMAs whose D layout (lane: one column, four consecutive rows) IS the A-operand
// layout of the 16x16x16 MFMA that consumes P^T / dS^T / dS next - no cross-lane movement; the B operands of those
// products have the reduction index as their row index in memory and come out of the row-major LDS tiles through
// ds_read_b64_tr_b16.
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

namespace {

// Tile rows of 288 bytes: the 8 rows x 32 bytes of a ds_read_b64_tr_b16 phase land on 8 different bank groups (272-byte rows -
// conflict-free for the b128 fragment reads instead - left every transpose read 2-way conflicted: 469 -> 452 us per C2 decoder
// layer; an XOR-swizzled unpadded layout that frees BOTH kinds of read measured 495: its 12 per-lane offsets cost more address
// arithmetic and registers than the conflicts it removes).
constexpr int D = 128, TB = 64, LDT = D + 16;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

__device__ inline s16x4_t tr4(const unsigned short* a) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)a); }
__device__ inline s16x4_t pack4(const float* f) {
    union { uint2 u; s16x4_t v; } c;
    c.u = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    return c.v;
}
__device__ inline f32x4_t mma16(s16x4_t a, s16x4_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// 64 rows x 128 columns of a (rows, ld) bf16 matrix: global -> registers (issued a whole block ahead of its use) -> LDS tile;
// rows >= nrows are zero
__device__ inline void fetch_tile(uint4 (&x)[4], const unsigned short* g, long ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + 256 * i, r = v >> 4, c = (v & 15) * 8;
        x[i] = row0 + r < nrows ? *(const uint4*)(g + (long)(row0 + r) * ld + c) : make_uint4(0, 0, 0, 0);
    }
}
__device__ inline void stash_tile(unsigned short* s, const uint4 (&x)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + 256 * i;
        *(uint4*)(s + (v >> 4) * LDT + (v & 15) * 8) = x[i];
    }
}
// this lane's four 16-byte fragments (k = ks*32 + fg*8 .. +7) of row `row` of a (rows, ld) matrix, zero past nrows
__device__ inline void load_frags(uint4 (&f)[4], const unsigned short* g, long ld, int row, int nrows, int fg) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = row < nrows ? *(const uint4*)(g + (long)row * ld + ks * 32 + fg * 8) : make_uint4(0, 0, 0, 0);
}

// NB = 16-key (dK/dV) or 16-query (dQ) blocks per wave: 2 halves the LDS fragment / transpose reads per MFMA and the number of
// times the walked tiles are fetched, for 2x the accumulators
// DROP: the attention-weight dropout of the forward is regenerated (a run-time test put four branches into every block epilogue)
template <int NB, bool DROP>
__global__ __launch_bounds__(256, NB <= 2 ? 2 : 1) void attn_bwd_dkdv_kernel(AttnBwdArgs p) {
    // two buffers: the next block's tiles are written while this one is multiplied - one barrier per block instead of two
    __shared__ __attribute__((aligned(16))) unsigned short sQ2[2][TB * LDT], sO2[2][TB * LDT];
    __shared__ __attribute__((aligned(16))) float sL2[2][TB], sD2[2][TB];  // sL = +inf past the last query: exp2(.. - inf) = 0, no select
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh % p.heads;
    const int k0 = blockIdx.x * (TB * NB);
    const long ldq = 3L * p.H;
    const unsigned short* Q = (const unsigned short*)p.qkv + (long)b * p.S * ldq + h * D;
    const unsigned short* dO = (const unsigned short*)p.dout + (long)b * p.S * p.H + h * D;
    int key[NB];       // the key of this lane's column in the S / dP blocks
    bool kvalid[NB];
    uint4 Kf[NB][4], Vf[NB][4];
    f32x4_t dK[NB][8], dV[NB][8];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        key[kb] = k0 + (16 * NB) * w + 16 * kb + fr;
        kvalid[kb] = key[kb] < p.S && !(p.key_pad && p.key_pad[(long)b * p.S + key[kb]]);
        load_frags(Kf[kb], Q + p.H, ldq, key[kb], p.S, fg);
        load_frags(Vf[kb], Q + 2 * p.H, ldq, key[kb], p.S, fg);
#pragma unroll
        for (int i = 0; i < 8; ++i) dK[kb][i] = dV[kb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const float* lse = p.lse2 + (long)bh * p.S;
    const float* dl = p.delta + (long)bh * p.S;
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
    const float dsc = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;

    uint4 xq[4], xo[4];
    float xl = 0.f, xd = 0.f;
    fetch_tile(xq, Q, ldq, 0, p.S);
    fetch_tile(xo, dO, p.H, 0, p.S);
    if (tid < TB) { xl = tid < p.S ? lse[tid] : __builtin_inff(); xd = tid < p.S ? dl[tid] : 0.f; }
    stash_tile(sQ2[0], xq);
    stash_tile(sO2[0], xo);
    if (tid < TB) { sL2[0][tid] = xl; sD2[0][tid] = xd; }
    __syncthreads();
    for (int q0 = 0, cur = 0; q0 < p.S; q0 += TB, cur ^= 1) {
        const unsigned short* sQ = sQ2[cur];
        const unsigned short* sO = sO2[cur];
        const float* sL = sL2[cur];
        const float* sD = sD2[cur];
        const bool more = q0 + TB < p.S;
        if (more) {  // the next block's loads fly while this one is computed
            fetch_tile(xq, Q, ldq, q0 + TB, p.S);
            fetch_tile(xo, dO, p.H, q0 + TB, p.S);
            if (tid < TB) { xl = q0 + TB + tid < p.S ? lse[q0 + TB + tid] : __builtin_inff(); xd = q0 + TB + tid < p.S ? dl[q0 + TB + tid] : 0.f; }
        }
#pragma unroll 1
        for (int qp = 0; qp < 2; ++qp) {  // 32 queries per step: two 16-query S / dP blocks feed ONE 16x16x32 product each for dV and dK
            f32x4_t s[2][NB], dp[2][NB];
            float lq[2][4], dq4[2][4];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                // this lane's four queries' statistics in two 16-byte reads, issued ahead of the products (they were four serialised
                // LDS round trips inside exec-masked branches)
                const float4 l4 = *(const float4*)(sL + qp * 32 + hb * 16 + fg * 4), d4 = *(const float4*)(sD + qp * 32 + hb * 16 + fg * 4);
                lq[hb][0] = l4.x; lq[hb][1] = l4.y; lq[hb][2] = l4.z; lq[hb][3] = l4.w;
                dq4[hb][0] = d4.x; dq4[hb][1] = d4.y; dq4[hb][2] = d4.z; dq4[hb][3] = d4.w;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) s[hb][kb] = dp[hb][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const uint4 qa = *(const uint4*)(sQ + (qp * 32 + hb * 16 + fr) * LDT + ks * 32 + fg * 8);
                    const uint4 oa = *(const uint4*)(sO + (qp * 32 + hb * 16 + fr) * LDT + ks * 32 + fg * 8);
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) {
                        Mma16<bf16>::step(qa, Kf[kb][ks], s[hb][kb]);    // S[q = fg*4 + r][key = fr]
                        Mma16<bf16>::step(oa, Vf[kb][ks], dp[hb][kb]);   // dP, same layout
                    }
                }
            // A operands of the 16x16x32 products: row = key (fr), reduction slot fg*8 + j <-> query fg*4 + j of the first block
            // (j < 4) / fg*4 + j - 4 of the second: the D layout of the two S blocks, side by side - no cross-lane movement; the B
            // operands pair the same queries (two transpose reads, one per block)
            union AB { s16x4_t h[2]; bf16x8_t v; };
            AB pa[NB], da[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    float pv[4], dsv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[hb][kb][r], p.scale_log2e, -lq[hb][r]));
                        const float pr = kvalid[kb] ? e : 0.f;
                        float dpe = dp[hb][kb][r], pd = pr;
                        if constexpr (DROP) {
                            const int q = q0 + qp * 32 + hb * 16 + fg * 4 + r;
                            const bool keep = dropout_bits(p.drop_seed, p.drop_key, ((uint64_t)bh * p.S + q) * p.S + key[kb]) >= thr;
                            pd = keep ? pr * dsc : 0.f;
                            dpe = keep ? dpe * dsc : 0.f;
                        }
                        pv[r] = pd;
                        dsv[r] = (pr * p.scale) * (dpe - dq4[hb][r]);
                    }
                    pa[kb].h[hb] = pack4(pv);
                    da[kb].h[hb] = pack4(dsv);
                }
            const unsigned short* tq = sQ + (qp * 32 + fg * 4 + (fr >> 2)) * LDT + (fr & 3) * 4;
            const unsigned short* to = sO + (qp * 32 + fg * 4 + (fr >> 2)) * LDT + (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                AB bo, bq;
                bo.h[0] = tr4(to + dt * 16); bo.h[1] = tr4(to + 16 * LDT + dt * 16);
                bq.h[0] = tr4(tq + dt * 16); bq.h[1] = tr4(tq + 16 * LDT + dt * 16);
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                    dV[kb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[kb].v, bo.v, dV[kb][dt], 0, 0, 0);   // += P^T dO
                    dK[kb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[kb].v, bq.v, dK[kb][dt], 0, 0, 0);   // += dS^T Q
                }
            }
        }
        if (more) {  // the other buffer was last read before the previous barrier
            stash_tile(sQ2[cur ^ 1], xq);
            stash_tile(sO2[cur ^ 1], xo);
            if (tid < TB) { sL2[cur ^ 1][tid] = xl; sD2[cur ^ 1][tid] = xd; }
        }
        __syncthreads();
    }
    // D: lane holds column d = dt*16 + fr, rows key = fg*4 + r of each 16-key block
    unsigned short* out = (unsigned short*)p.dqkv + (long)b * p.S * ldq + h * D;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = k0 + (16 * NB) * w + 16 * kb + fg * 4 + r;
                if (kk < p.S) {
                    out[(long)kk * ldq + p.H + dt * 16 + fr] = Num<bf16>::from_f32(dK[kb][dt][r]).v;
                    out[(long)kk * ldq + 2 * p.H + dt * 16 + fr] = Num<bf16>::from_f32(dV[kb][dt][r]).v;
                }
            }
}

template <int NB, bool DROP>
__global__ __launch_bounds__(256, NB == 1 ? 3 : (NB == 2 ? 2 : 1)) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short sK[TB * LDT], sV[TB * LDT];
    __shared__ __attribute__((aligned(16))) float sOk[TB];  // 0 for a key that takes part, -inf otherwise (added to the exponent)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh % p.heads;
    const int q0 = blockIdx.x * (TB * NB);
    const long ldq = 3L * p.H;
    const unsigned short* Q = (const unsigned short*)p.qkv + (long)b * p.S * ldq + h * D;
    const unsigned short* dO = (const unsigned short*)p.dout + (long)b * p.S * p.H + h * D;
    int q[NB];  // the query of this lane's column in the S^T / dP^T blocks
    bool qvalid[NB];
    uint4 Qf[NB][4], Of[NB][4];
    float lse[NB], dl[NB];
    f32x4_t dQ[NB][8];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        q[qb] = q0 + (16 * NB) * w + 16 * qb + fr;
        qvalid[qb] = q[qb] < p.S;
        load_frags(Qf[qb], Q, ldq, q[qb], p.S, fg);
        load_frags(Of[qb], dO, p.H, q[qb], p.S, fg);
        lse[qb] = qvalid[qb] ? p.lse2[(long)bh * p.S + q[qb]] : __builtin_inff();  // exp2(.. - inf) = 0: no select per element
        dl[qb] = qvalid[qb] ? p.delta[(long)bh * p.S + q[qb]] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dQ[qb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
    const float dsc = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;

    uint4 xk[4], xv[4];
    float xok = 0.f;
    fetch_tile(xk, Q + p.H, ldq, 0, p.S);
    fetch_tile(xv, Q + 2 * p.H, ldq, 0, p.S);
    if (tid < TB) xok = (tid < p.S && !(p.key_pad && p.key_pad[(long)b * p.S + tid])) ? 0.f : -__builtin_inff();
    for (int k0 = 0; k0 < p.S; k0 += TB) {
        __syncthreads();
        stash_tile(sK, xk);
        stash_tile(sV, xv);
        if (tid < TB) sOk[tid] = xok;
        __syncthreads();
        if (k0 + TB < p.S) {
            fetch_tile(xk, Q + p.H, ldq, k0 + TB, p.S);
            fetch_tile(xv, Q + 2 * p.H, ldq, k0 + TB, p.S);
            if (tid < TB) xok = (k0 + TB + tid < p.S && !(p.key_pad && p.key_pad[(long)b * p.S + k0 + TB + tid])) ? 0.f : -__builtin_inff();
        }
#pragma unroll 1
        for (int kp = 0; kp < 2; ++kp) {  // 32 keys per step: two 16-key S^T / dP^T blocks feed one 16x16x32 product for dQ
            f32x4_t s[2][NB], dp[2][NB];
            float ok4[2][4];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const float4 o4 = *(const float4*)(sOk + kp * 32 + hb * 16 + fg * 4);  // this lane's four keys, one read ahead of the products
                ok4[hb][0] = o4.x; ok4[hb][1] = o4.y; ok4[hb][2] = o4.z; ok4[hb][3] = o4.w;
#pragma unroll
                for (int qb = 0; qb < NB; ++qb) s[hb][qb] = dp[hb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const uint4 ka = *(const uint4*)(sK + (kp * 32 + hb * 16 + fr) * LDT + ks * 32 + fg * 8);
                    const uint4 va = *(const uint4*)(sV + (kp * 32 + hb * 16 + fr) * LDT + ks * 32 + fg * 8);
#pragma unroll
                    for (int qb = 0; qb < NB; ++qb) {
                        Mma16<bf16>::step(ka, Qf[qb][ks], s[hb][qb]);    // S^T[key = fg*4 + r][q = fr]
                        Mma16<bf16>::step(va, Of[qb][ks], dp[hb][qb]);   // dP^T
                    }
                }
            union AB { s16x4_t h[2]; bf16x8_t v; };
            AB da[NB];  // A operand: row = query (fr), reduction slot fg*8 + j <-> key fg*4 + j of the first block / fg*4 + j - 4 of the second
#pragma unroll
            for (int qb = 0; qb < NB; ++qb)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    float dsv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[hb][qb][r], p.scale_log2e, ok4[hb][r] - lse[qb]));
                        float dpe = dp[hb][qb][r];
                        if constexpr (DROP) {
                            const int key = k0 + kp * 32 + hb * 16 + fg * 4 + r;
                            dpe = dropout_bits(p.drop_seed, p.drop_key, ((uint64_t)bh * p.S + q[qb]) * p.S + key) >= thr ? dpe * dsc : 0.f;
                        }
                        dsv[r] = (pr * p.scale) * (dpe - dl[qb]);
                    }
                    da[qb].h[hb] = pack4(dsv);
                }
            const unsigned short* tk = sK + (kp * 32 + fg * 4 + (fr >> 2)) * LDT + (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                AB bk;
                bk.h[0] = tr4(tk + dt * 16); bk.h[1] = tr4(tk + 16 * LDT + dt * 16);
#pragma unroll
                for (int qb = 0; qb < NB; ++qb) dQ[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[qb].v, bk.v, dQ[qb][dt], 0, 0, 0);  // += dS K
            }
        }
    }
    unsigned short* out = (unsigned short*)p.dqkv + (long)b * p.S * ldq + h * D;
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + (16 * NB) * w + 16 * qb + fg * 4 + r;
                if (qq < p.S) out[(long)qq * ldq + dt * 16 + fr] = Num<bf16>::from_f32(dQ[qb][dt][r]).v;
            }
}

// 16-row blocks per wave, per launch (tools/bench_ops.py flash, PMC in profiles/r02_pmc_attention_bwd.md): the dK / dV launch
// stays at 1 (2 doubles its 128 accumulator registers and spills at two waves per SIMD), the dQ launch takes 2 (half the LDS
// fragment / transpose reads per MFMA, 234 registers, no spill: MFMA pipe 45 -> 51 % busy)
// r03: 3 / 4 blocks per wave (48 / 64 keys or queries per wave, ONE wave per SIMD with the 512-register file; dK dV at 3: 456
// registers, no spill, a third of the LDS fragment / transpose reads per MFMA; at 4: 117 spilled) behind knobs 905-908.  On random
// operands (tools/bench_ops.py flash) the C2 decoder pair goes 370 -> 329 us with the 3-block dK / dV launch; inside the training
// step, on the model's own activations, that launch takes 221 us against 219 us at 1 block (rocprof, same box, both ways) - the
// gain does not survive real data (the launch is power / clock limited there), so the default stays 1.
// (Tuning::attn_bwd_nb / attn_bwd_nb_dq; dQ: 0 = by size - 2 when that still leaves two workgroups per CU: C2 encoder 24 us at 1, 28 at 2)

}  // namespace

bool attention_bwd_supported(int dtype, int H, int heads) { return dtype == FS2_BF16 && heads > 0 && H == heads * D; }

int launch_attention_bwd(const AttnBwdArgs& a, int dtype, hipStream_t stream) {
    if (!attention_bwd_supported(dtype, a.H, a.heads) || a.B <= 0 || a.S <= 0) return FS2_ERR_SHAPE;
    if (!a.qkv || !a.dout || !a.lse2 || !a.delta || !a.dqkv) return FS2_ERR_ARG;
    const bool drop = a.drop_p > 0.f;
#define FS2_AB(KERNEL, NBV, DR) \
    hipLaunchKernelGGL((KERNEL<NBV, DR>), dim3((a.S + NBV * TB - 1) / (NBV * TB), a.B * a.heads), dim3(256), 0, stream, a)
    const Tuning& tn = tuning_of(a.tune);
    const int nb_kv = tn.attn_bwd_nb;
    if (nb_kv == 4 && !drop) FS2_AB(attn_bwd_dkdv_kernel, 4, false);
    else if (nb_kv == 3 && !drop) FS2_AB(attn_bwd_dkdv_kernel, 3, false);
    else if (nb_kv == 2) { if (drop) FS2_AB(attn_bwd_dkdv_kernel, 2, true); else FS2_AB(attn_bwd_dkdv_kernel, 2, false); }
    else { if (drop) FS2_AB(attn_bwd_dkdv_kernel, 1, true); else FS2_AB(attn_bwd_dkdv_kernel, 1, false); }
    const int nb_dq = tn.attn_bwd_nb_dq ? tn.attn_bwd_nb_dq : ((long)((a.S + 2 * TB - 1) / (2 * TB)) * a.B * a.heads >= 512 ? 2 : 1);
    if (nb_dq == 4 && !drop) FS2_AB(attn_bwd_dq_kernel, 4, false);
    else if (nb_dq == 3 && !drop) FS2_AB(attn_bwd_dq_kernel, 3, false);
    else if (nb_dq >= 2) { if (drop) FS2_AB(attn_bwd_dq_kernel, 2, true); else FS2_AB(attn_bwd_dq_kernel, 2, false); }
    else { if (drop) FS2_AB(attn_bwd_dq_kernel, 1, true); else FS2_AB(attn_bwd_dq_kernel, 1, false); }
#undef FS2_AB
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
