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

constexpr int D = 128, TB = 64, LDT = D + 8;  // tile rows of 272 bytes: b128 fragment reads conflict-free
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

__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnBwdArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short sQ[TB * LDT], sO[TB * LDT];
    __shared__ float sL[TB], sD[TB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh % p.heads;
    const int k0 = blockIdx.x * TB;
    const long ldq = 3L * p.H;
    const unsigned short* Q = (const unsigned short*)p.qkv + (long)b * p.S * ldq + h * D;
    const unsigned short* dO = (const unsigned short*)p.dout + (long)b * p.S * p.H + h * D;
    const int key = k0 + 16 * w + fr;  // the key of this lane's column in the S / dP blocks
    const bool kvalid = key < p.S && !(p.key_pad && p.key_pad[(long)b * p.S + key]);
    uint4 Kf[4], Vf[4];
    load_frags(Kf, Q + p.H, ldq, key, p.S, fg);
    load_frags(Vf, Q + 2 * p.H, ldq, key, p.S, fg);
    f32x4_t dK[8], dV[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dK[i] = dV[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float* lse = p.lse2 + (long)bh * p.S;
    const float* dl = p.delta + (long)bh * p.S;
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
    const float dsc = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;

    uint4 xq[4], xo[4];
    float xl = 0.f, xd = 0.f;
    fetch_tile(xq, Q, ldq, 0, p.S);
    fetch_tile(xo, dO, p.H, 0, p.S);
    if (tid < TB) { xl = tid < p.S ? lse[tid] : 0.f; xd = tid < p.S ? dl[tid] : 0.f; }
    for (int q0 = 0; q0 < p.S; q0 += TB) {
        __syncthreads();  // everyone is done with the previous tiles
        stash_tile(sQ, xq);
        stash_tile(sO, xo);
        if (tid < TB) { sL[tid] = xl; sD[tid] = xd; }
        __syncthreads();
        if (q0 + TB < p.S) {  // the next block's loads fly while this one is computed
            fetch_tile(xq, Q, ldq, q0 + TB, p.S);
            fetch_tile(xo, dO, p.H, q0 + TB, p.S);
            if (tid < TB) { xl = q0 + TB + tid < p.S ? lse[q0 + TB + tid] : 0.f; xd = q0 + TB + tid < p.S ? dl[q0 + TB + tid] : 0.f; }
        }
#pragma unroll 1  // unrolling by 2 costs 24 VGPRs and a wave per SIMD: 323 -> 515 us
        for (int qs = 0; qs < 4; ++qs) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 qa = *(const uint4*)(sQ + (qs * 16 + fr) * LDT + ks * 32 + fg * 8);
                const uint4 oa = *(const uint4*)(sO + (qs * 16 + fr) * LDT + ks * 32 + fg * 8);
                Mma16<bf16>::step(qa, Kf[ks], s);    // S[q = fg*4 + r][key = fr]
                Mma16<bf16>::step(oa, Vf[ks], dp);   // dP, same layout
            }
            float pv[4], dsv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qs * 16 + fg * 4 + r, q = q0 + ql;
                float pr = (kvalid && q < p.S) ? __builtin_amdgcn_exp2f(s[r] * p.scale_log2e - sL[ql]) : 0.f;
                float dpe = dp[r];
                float pd = pr;
                if (p.drop_p > 0.f) {
                    const bool keep = dropout_bits(p.drop_seed, p.drop_key, ((uint64_t)bh * p.S + q) * p.S + key) >= thr;
                    pd = keep ? pr * dsc : 0.f;
                    dpe = keep ? dpe * dsc : 0.f;
                }
                pv[r] = pd;
                dsv[r] = pr * (dpe - sD[ql]) * p.scale;
            }
            const s16x4_t pa = pack4(pv), da = pack4(dsv);  // A operands: row = key (fr), k = the four queries fg*4 ..
            const unsigned short* tq = sQ + (qs * 16 + fg * 4 + (fr >> 2)) * LDT + (fr & 3) * 4;
            const unsigned short* to = sO + (qs * 16 + fg * 4 + (fr >> 2)) * LDT + (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                dV[dt] = mma16(pa, tr4(to + dt * 16), dV[dt]);   // += P^T dO
                dK[dt] = mma16(da, tr4(tq + dt * 16), dK[dt]);   // += dS^T Q
            }
        }
    }
    // D: lane holds column d = dt*16 + fr, rows key = fg*4 + r of this wave's 16 keys
    unsigned short* out = (unsigned short*)p.dqkv + (long)b * p.S * ldq + h * D;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = k0 + 16 * w + fg * 4 + r;
            if (kk < p.S) {
                out[(long)kk * ldq + p.H + dt * 16 + fr] = Num<bf16>::from_f32(dK[dt][r]).v;
                out[(long)kk * ldq + 2 * p.H + dt * 16 + fr] = Num<bf16>::from_f32(dV[dt][r]).v;
            }
        }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short sK[TB * LDT], sV[TB * LDT];
    __shared__ unsigned char sOk[TB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh % p.heads;
    const int q0 = blockIdx.x * TB;
    const long ldq = 3L * p.H;
    const unsigned short* Q = (const unsigned short*)p.qkv + (long)b * p.S * ldq + h * D;
    const unsigned short* dO = (const unsigned short*)p.dout + (long)b * p.S * p.H + h * D;
    const int q = q0 + 16 * w + fr;  // the query of this lane's column in the S^T / dP^T blocks
    const bool qvalid = q < p.S;
    uint4 Qf[4], Of[4];
    load_frags(Qf, Q, ldq, q, p.S, fg);
    load_frags(Of, dO, p.H, q, p.S, fg);
    const float lse = qvalid ? p.lse2[(long)bh * p.S + q] : 0.f;
    const float dl = qvalid ? p.delta[(long)bh * p.S + q] : 0.f;
    f32x4_t dQ[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dQ[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
    const float dsc = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;

    uint4 xk[4], xv[4];
    unsigned char xok = 0;
    fetch_tile(xk, Q + p.H, ldq, 0, p.S);
    fetch_tile(xv, Q + 2 * p.H, ldq, 0, p.S);
    if (tid < TB) xok = (tid < p.S && !(p.key_pad && p.key_pad[(long)b * p.S + tid])) ? 1 : 0;
    for (int k0 = 0; k0 < p.S; k0 += TB) {
        __syncthreads();
        stash_tile(sK, xk);
        stash_tile(sV, xv);
        if (tid < TB) sOk[tid] = xok;
        __syncthreads();
        if (k0 + TB < p.S) {
            fetch_tile(xk, Q + p.H, ldq, k0 + TB, p.S);
            fetch_tile(xv, Q + 2 * p.H, ldq, k0 + TB, p.S);
            if (tid < TB) xok = (k0 + TB + tid < p.S && !(p.key_pad && p.key_pad[(long)b * p.S + k0 + TB + tid])) ? 1 : 0;
        }
#pragma unroll 2
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 ka = *(const uint4*)(sK + (ks4 * 16 + fr) * LDT + ks * 32 + fg * 8);
                const uint4 va = *(const uint4*)(sV + (ks4 * 16 + fr) * LDT + ks * 32 + fg * 8);
                Mma16<bf16>::step(ka, Qf[ks], s);    // S^T[key = fg*4 + r][q = fr]
                Mma16<bf16>::step(va, Of[ks], dp);   // dP^T
            }
            float dsv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = ks4 * 16 + fg * 4 + r, key = k0 + kl;
                const float pr = (qvalid && sOk[kl]) ? __builtin_amdgcn_exp2f(s[r] * p.scale_log2e - lse) : 0.f;
                float dpe = dp[r];
                if (p.drop_p > 0.f)
                    dpe = dropout_bits(p.drop_seed, p.drop_key, ((uint64_t)bh * p.S + q) * p.S + key) >= thr ? dpe * dsc : 0.f;
                dsv[r] = pr * (dpe - dl) * p.scale;
            }
            const s16x4_t da = pack4(dsv);  // A operand: row = query (fr), k = the four keys fg*4 ..
            const unsigned short* tk = sK + (ks4 * 16 + fg * 4 + (fr >> 2)) * LDT + (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) dQ[dt] = mma16(da, tr4(tk + dt * 16), dQ[dt]);  // += dS K
        }
    }
    unsigned short* out = (unsigned short*)p.dqkv + (long)b * p.S * ldq + h * D;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q0 + 16 * w + fg * 4 + r;
            if (qq < p.S) out[(long)qq * ldq + dt * 16 + fr] = Num<bf16>::from_f32(dQ[dt][r]).v;
        }
}

}  // namespace

bool attention_bwd_supported(int dtype, int H, int heads) { return dtype == FS2_BF16 && heads > 0 && H == heads * D; }

int launch_attention_bwd(const AttnBwdArgs& a, int dtype, hipStream_t stream) {
    if (!attention_bwd_supported(dtype, a.H, a.heads) || a.B <= 0 || a.S <= 0) return FS2_ERR_SHAPE;
    if (!a.qkv || !a.dout || !a.lse2 || !a.delta || !a.dqkv) return FS2_ERR_ARG;
    const dim3 grid((a.S + TB - 1) / TB, a.B * a.heads);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
