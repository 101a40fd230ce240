// General strided-batched GEMM for the training step's backward:
//   C[b][m][n] = alpha * sum_k A[b](m, k) * B[b](k, n)  (+ bias[n])  (+ beta * C[b][m][n])
// with every operand addressed by element strides, so one kernel covers
//   NT  (both operands k-contiguous)        attention scores  S = Q K^T,   dP = dO V^T
//   NN  (B n-contiguous)                    linear / conv dgrad  dX = dY W,  O = P V,  dQ = dS K
//   TN  (both operands k-strided)           weight gradients  dW = dY^T X,  dV = P^T dO,  dK = dS^T Q
// plus the two implicit-conv forms of a 'same'-padded Conv1d over (B*S, C) time-major activations:
//   dgrad: K = taps * Kin, k-tile `tap` reads A rows m + a_shift0 + tap * a_shift_step (zero outside the row's
//          utterance of `seg` rows) and B from + tap * sBtap;
//   wgrad: batch index b2 = tap, B's k index (a time row) is read at k + b_shift0 + b2 * b_shift_step, zero outside
//          the utterance.
// fp32 operands on the exact fp32 MFMA (16x16x4), fp32 accumulation; 128 x 128 x 16 tiles, 4 waves (2 x 2),
// register-prefetched double-buffered LDS.  Split-K (deterministic: per-split slabs + a reduce pass) for the
// weight gradients, whose reduction runs over all B*T rows while the output is only N x K.
// The reference gets all of these from autograd (fastspeech2.py:786-797 training_step -> loss.backward()).
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDK = 20;    // k-contiguous LDS image: [128][20] floats (16-B aligned rows)
constexpr int LDM = 144;   // m/n-contiguous LDS image: [16][144] floats (rows 16 banks apart)
constexpr int OPSZ = BM * LDK;  // 2560 floats >= BK * LDM = 2304

// dS = alpha * P * (dropout(dP) - delta[row]) for one element of the dP = dO V^T product (alpha = 1 / sqrt(d); the identity
// sum_k dP P = sum_d dO O = delta holds with the attention-weight dropout in place because O was formed from the dropped P)
template <typename OutT>
__device__ inline float softmax_bwd_epilogue(const BGemmArgs& p, float dp, int b1, int b2, int m, int n) {
    const long bidx = (long)b1 * p.nb2 + b2;
    if (p.drop_p > 0.f) {
        const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
        dp = dropout_bits(p.drop_seed, p.drop_key, ((uint64_t)bidx * p.M + m) * p.N + n) >= thr ? dp * (1.f / (1.f - p.drop_p)) : 0.f;
    }
    const float pv = Num<OutT>::to_f32(((const OutT*)p.epi_p)[b1 * p.sC1 + b2 * p.sC2 + (long)m * p.ldc + n]);
    return p.alpha * pv * (dp - p.epi_delta[bidx * p.M + m]);
}

struct Operand {
    const float* p;
    long s_mn, s_k;  // element strides of the tile's outer (m or n) index and of k
    int mn0, MN;     // tile origin and extent of the outer index
    int shift_mn;    // rows of the outer index are read at +shift (dgrad), zero outside the segment
    int shift_k;     // k rows are read at +shift (wgrad)
    int seg;
    bool vec;
};

// One tile (128 outer x 16 k) -> two float4 per thread.  kc: vectors run along k; else along the outer index.
__device__ inline void load_tile(const Operand& o, int k0, int Kend, float4 (&r)[2]) {
    const int tid = threadIdx.x;
    const bool kc = o.s_k == 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + 256 * i;
        int mn, k;
        if (kc) { mn = v >> 2; k = (v & 3) * 4; } else { k = v >> 5; mn = (v & 31) * 4; }
        const int gmn = o.mn0 + mn, gk = k0 + k;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (kc) {
            bool ok = gmn < o.MN;
            long row = gmn;
            if (o.seg && ok) {
                const int in = gmn % o.seg + o.shift_mn;
                ok = in >= 0 && in < o.seg;
                row = gmn + o.shift_mn;
            }
            if (ok) {
                const float* src = o.p + row * o.s_mn + gk;
                if (o.vec && gk + 3 < Kend) {
                    const float4 t = *(const float4*)src;
                    f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gk + e < Kend) f[e] = src[e];
                }
            }
        } else {
            bool ok = gk < Kend;
            long krow = gk;
            if (o.seg && ok) {
                const int in = gk % o.seg + o.shift_k;
                ok = in >= 0 && in < o.seg;
                krow = gk + o.shift_k;
            }
            if (ok) {
                const float* src = o.p + krow * o.s_k + gmn;
                if (o.vec && gmn + 3 < o.MN) {
                    const float4 t = *(const float4*)src;
                    f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gmn + e < o.MN) f[e] = src[e];
                }
            }
        }
        r[i] = make_float4(f[0], f[1], f[2], f[3]);
    }
}

__device__ inline void store_tile(float* s, bool kc, const float4 (&r)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + 256 * i;
        if (kc) *(float4*)(s + (v >> 2) * LDK + (v & 3) * 4) = r[i];
        else *(float4*)(s + (v >> 5) * LDM + (v & 31) * 4) = r[i];
    }
}

template <bool AKC, bool BKC>  // operand layouts at compile time: the fragment reads are straight-line code (see the bf16 kernel)
__global__ __launch_bounds__(256) void bgemm_f32_kernel(BGemmArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][OPSZ];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int z = blockIdx.z;
    const int split = z % splitk;
    z /= splitk;
    const int b2 = z % p.nb2, b1 = z / p.nb2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    Operand A, B;
    A.p = (const float*)p.A + b1 * p.sA1 + b2 * p.sA2;
    A.s_mn = p.sAm; A.s_k = p.sAk; A.mn0 = m0; A.MN = p.M;
    A.shift_mn = 0; A.shift_k = 0; A.seg = 0; A.vec = p.vecA;
    B.p = (const float*)p.B + b1 * p.sB1 + b2 * p.sB2;
    B.s_mn = p.sBn; B.s_k = p.sBk; B.mn0 = n0; B.MN = p.N;
    B.shift_mn = 0; B.shift_k = 0; B.seg = 0; B.vec = p.vecB;
    if (p.seg && p.taps <= 1) {  // wgrad form: B's k rows shifted by the tap this batch index stands for
        B.seg = p.seg;
        B.shift_k = p.b_shift0 + b2 * p.b_shift_step;
    }
    constexpr bool akc = AKC, bkc = BKC;

    // k range of this split, in whole k-tiles; in the dgrad form K = taps * Kin and tiles never straddle a tap
    const int Kin = p.taps > 1 ? p.Kin : p.K;
    const int tiles_per_tap = (Kin + BK - 1) / BK;
    const int ntiles = tiles_per_tap * (p.taps > 1 ? p.taps : 1);
    const int per = (ntiles + splitk - 1) / splitk;
    const int t_begin = split * per, t_end = min(ntiles, t_begin + per);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const float* Ap0 = A.p;
    const float* Bp0 = B.p;
    auto fetch = [&](int t, float4 (&ra)[2], float4 (&rb)[2]) {
        int k0 = t * BK, kend = p.K;
        if (p.taps > 1) {
            const int tap = t / tiles_per_tap;
            k0 = (t - tap * tiles_per_tap) * BK;
            kend = Kin;
            A.seg = p.seg;
            A.shift_mn = p.a_shift0 + tap * p.a_shift_step;
            A.p = Ap0;
            B.p = Bp0 + tap * p.sBtap;
        }
        load_tile(A, k0, kend, ra);
        load_tile(B, k0, kend, rb);
    };

    float4 ra[2], rb[2];
    if (t_begin < t_end) {
        fetch(t_begin, ra, rb);
        store_tile(lds[0][0], akc, ra);
        store_tile(lds[0][1], bkc, rb);
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) fetch(t + 1, ra, rb);
        const float* sa = lds[cur][0];
        const float* sb = lds[cur][1];
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int k = ks * 4 + fg;
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = wm * 64 + i * 16 + fr;
                a[i] = akc ? sa[m * LDK + k] : sa[k * LDM + m];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = wn * 64 + j * 16 + fr;
                b[j] = bkc ? sb[n * LDK + k] : sb[k * LDM + n];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_tile(lds[cur ^ 1][0], akc, ra);
            store_tile(lds[cur ^ 1][1], bkc, rb);
        }
        __syncthreads();
    }

    // D: lane holds column fr, rows fg * 4 + r of each 16 x 16 block
    if (splitk > 1) {
        float* ws = p.ws + ((long)(b1 * p.nb2 + b2) * splitk + split) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + fg * 4 + r, n = n0 + wn * 64 + j * 16 + fr;
                    if (m < p.M && n < p.N) ws[(long)m * p.N + n] = acc[i][j][r];
                }
        return;
    }
    float* C = (float*)p.C + b1 * p.sC1 + b2 * p.sC2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fr;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
                if (m >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bv;
                float* dst = C + (long)m * p.ldc + n;
                if (p.epi_p) v = softmax_bwd_epilogue<float>(p, acc[i][j][r], b1, b2, m, n);
                if (p.beta != 0.f) v += p.beta * *dst;
                *dst = v;
            }
        }
}

// C = alpha * sum over splits + bias + beta * C, one thread per output element
__global__ void bgemm_reduce_kernel(BGemmArgs p) {
    const long per = (long)p.M * p.N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int bz = blockIdx.y;  // batch
    if (i >= per) return;
    const int b2 = bz % p.nb2, b1 = bz / p.nb2;
    const float* ws = p.ws + (long)bz * p.splitk * per + i;
    float s = 0.f;
    for (int k = 0; k < p.splitk; ++k) s += ws[(long)k * per];
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float v = p.alpha * s + (p.bias ? p.bias[n] : 0.f);
    float* dst = (float*)p.C + b1 * p.sC1 + b2 * p.sC2 + (long)m * p.ldc + n;
    if (p.beta != 0.f) v += p.beta * *dst;
    *dst = v;
}


// ---- bf16 operands (the mixed-precision training path): MFMA 16x16x32, fp32 accumulation ---------------------------
// 128 x 128 x 32 tiles, 4 waves (2 x 2).  Each operand is copied to LDS in its memory order with 16-byte vectors:
//   k-contiguous operand  -> [128 rows][32 k] bf16, 80-byte rows, the row's four 16-byte chunks XOR-ed with (row / 8) & 3;
//                            fragments by ds_read_b128;
//   outer-contiguous operand (the k-strided side of NN / TN products) -> [32 k][128 outer] bf16, 288-byte rows; fragments by
//                            two ds_read_b64_tr_b16 each: inside a 16-lane group lane i hands in the address of
//                            [k0 + i/4][col0 + 4*(i%4)] and receives column col0 + i of that 4 x 16 block
//                            (tools/probes/tr_read_probe.hip), i.e. four consecutive k of its own MFMA row / column.
constexpr int HBK = 32, HLD = 40;
template <typename T> __device__ inline void ld4(const T* p, float* f);
template <> __device__ inline void ld4<float>(const float* p, float* f) {
    const float4 v = *(const float4*)p;
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ inline void ld4<bf16>(const bf16* p, float* f) {
    const uint2 v = *(const uint2*)p;
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ inline void st4(T* p, const float* f);
template <> __device__ inline void st4<float>(float* p, const float* f) { *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); }
template <> __device__ inline void st4<bf16>(bf16* p, const float* f) {
    *(uint2*)p = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
}
constexpr int HOPSZ = BM * HLD;  // bf16 elements per operand buffer (10 KB; the transposed image needs 32 * 144)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct OperandH {
    const unsigned short* p;
    long s_mn, s_k;
    int mn0, MN, shift_mn, shift_k, seg;
    bool vec;
};

__device__ inline uint4 load8(const unsigned short* src, bool vec, int nvalid) {  // nvalid of 8 elements in range
    if (vec && nvalid >= 8) return *(const uint4*)src;
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = i < nvalid ? src[i] : (unsigned short)0;
    return make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16),
                      e[6] | ((uint32_t)e[7] << 16));
}

// R = rows of the operand's outer index in the tile (128, or 256 for the A side of the 256-row tile)
template <int R, bool KC>
__device__ inline void load_tile_h(const OperandH& o, int k0, int Kend, uint4 (&r)[R / 64]) {
    const int tid = threadIdx.x;
    if (KC) {  // k-contiguous: (row, 8-k chunk)
#pragma unroll
        for (int i = 0; i < R / 64; ++i) {
            const int v = tid + 256 * i, mn = v >> 2, gk = k0 + (v & 3) * 8;
            const int gmn = o.mn0 + mn;
            bool ok = gmn < o.MN && gk < Kend;
            long row = gmn;
            if (o.seg && ok) {
                const int in = gmn % o.seg + o.shift_mn;
                ok = in >= 0 && in < o.seg;
                row = gmn + o.shift_mn;
            }
            r[i] = ok ? load8(o.p + row * o.s_mn + gk, o.vec, Kend - gk) : make_uint4(0, 0, 0, 0);
        }
    } else {  // outer-contiguous: (k row, 8 outer positions)
        constexpr int VR = R / 8;  // vectors per k row
#pragma unroll
        for (int i = 0; i < R / 64; ++i) {
            const int v = tid + 256 * i, gk = k0 + v / VR, gmn = o.mn0 + (v % VR) * 8;
            bool ok = gk < Kend && gmn < o.MN;
            long krow = gk;
            if (o.seg && ok) {
                const int in = gk % o.seg + o.shift_k;
                ok = in >= 0 && in < o.seg;
                krow = gk + o.shift_k;
            }
            r[i] = ok ? load8(o.p + krow * o.s_k + gmn, o.vec, o.MN - gmn) : make_uint4(0, 0, 0, 0);
        }
    }
}

template <int R>
__device__ inline void store_tile_h(unsigned short* s, bool kc, const uint4 (&r)[R / 64]) {
    const int tid = threadIdx.x;
    if (kc) {
#pragma unroll
        for (int i = 0; i < R / 64; ++i) {
            const int v = tid + 256 * i, row = v >> 2, ch = v & 3;
            *(uint4*)(s + row * HLD + ((ch ^ ((row >> 3) & 3)) << 3)) = r[i];
        }
    } else {
        constexpr int VR = R / 8;
#pragma unroll
        for (int i = 0; i < R / 64; ++i) {
            const int v = tid + 256 * i;
            *(uint4*)(s + (v / VR) * (R + 16) + (v % VR) * 8) = r[i];
        }
    }
}

// Per-thread cursor over an operand's k-tiles for the plain (taps <= 1) forms: everything that does not change from one
// k-tile to the next (the thread's rows / columns, their bounds, the 64-bit addresses) is worked out once; a step is a
// pointer bump, a compare and the 16-byte load.  (The generic loader above spends ~10x the MFMA issue slots of a k-step on
// address arithmetic - integer modulo, 64-bit multiplies - and made the kernel VALU-bound.)
template <int R, bool KC>
struct TileCursor {
    static constexpr int NV = R / 64;
    const unsigned short* ptr[NV];
    int gk[NV];        // k index of the vector (kc: its first element; else: its k row)
    int pos[NV];       // outer-contiguous + seg: gk % seg
    int nmn[NV];       // valid elements along the vector's own direction that do not depend on k (outer-contiguous: MN - gmn)
    bool ok[NV];
    static constexpr bool kc = KC;
    bool vec;
    int seg, shift, Kend;
    long step;

    __device__ inline void init(const OperandH& o, int k0, int Kend_) {
        const int tid = threadIdx.x;
        vec = o.vec; seg = o.seg; shift = o.shift_k; Kend = Kend_;
        step = kc ? HBK : (long)HBK * o.s_k;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + 256 * i;
            if (kc) {
                const int gmn = o.mn0 + (v >> 2);
                gk[i] = k0 + (v & 3) * 8;
                ok[i] = gmn < o.MN;
                ptr[i] = o.p + (long)gmn * o.s_mn + gk[i];
                nmn[i] = 0; pos[i] = 0;
            } else {
                constexpr int VR = R / 8;
                const int gmn = o.mn0 + (v % VR) * 8;
                gk[i] = k0 + v / VR;
                ok[i] = gmn < o.MN;
                nmn[i] = o.MN - gmn;
                pos[i] = seg ? gk[i] % seg : 0;
                ptr[i] = o.p + ((long)gk[i] + shift) * o.s_k + gmn;
            }
        }
    }
    __device__ inline void load(uint4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            bool v = ok[i] && gk[i] < Kend;
            if (!kc && seg) v = v && (unsigned)(pos[i] + shift) < (unsigned)seg;
            r[i] = v ? load8(ptr[i], vec, kc ? Kend - gk[i] : nmn[i]) : make_uint4(0, 0, 0, 0);
        }
    }
    // full tiles of an aligned operand: unconditional 16-byte loads (no per-lane branch may sit around a load - hipcc waits
    // for it where the branch ends, i.e. before the MFMAs); a k row of another utterance (wgrad form) is read from a safe
    // address and zeroed by a select
    __device__ inline void load_full(uint4 (&r)[NV], const unsigned short* safe) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (!kc && seg) {
                const bool v = (unsigned)(pos[i] + shift) < (unsigned)seg;
                const uint4 x = *(const uint4*)(v ? ptr[i] : safe);
                r[i] = v ? x : make_uint4(0, 0, 0, 0);
            } else {
                r[i] = *(const uint4*)ptr[i];
            }
        }
    }
    __device__ inline void next() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            ptr[i] += step;
            gk[i] += HBK;
            if (!kc && seg) {
                pos[i] += HBK;
                while (pos[i] >= seg) pos[i] -= seg;
            }
        }
    }
};

// 8 consecutive k (k = 8*fg .. 8*fg + 7) of MFMA row / column `mn0 + fr`, from either LDS image (ldt = row stride of the
// transposed image)
__device__ inline uint4 frag_h(const unsigned short* s, bool kc, int mn0, int fr, int fg, int ldt) {
    if (kc) {
        const int row = mn0 + fr;
        return *(const uint4*)(s + row * HLD + ((fg ^ ((row >> 3) & 3)) << 3));
    }
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const unsigned short* a = s + (fg * 8 + (fr >> 2)) * ldt + mn0 + (fr & 3) * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)a);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a + 4 * ldt));
    union { s16x4_t v[2]; uint4 u; } c;
    c.v[0] = lo;
    c.v[1] = hi;
    return c.u;
}

// WM = 16-row fragments per wave along m: 4 -> the 128 x 128 tile in use (4 waves as 2 x 2, a wave owns 64 x 64).  WM = 8
// (256 x 128) was measured slower on every training shape - it drops the kernel to 1-2 waves per SIMD - and is not built.
// AKC / BKC: the operand is k-contiguous in memory (compile time: with run-time flags hipcc put a branch and a full
// s_waitcnt lgkmcnt(0) around every fragment read - eight serialised LDS round trips per k-step)
// FULL: M % tile == 0, N % 128 == 0, K % 32 == 0, aligned operands, plain form (taps <= 1): the loop then holds no bounds
// logic and no scalar fallback at all (the generic instantiation carries ~100 exec-mask branches and its SGPR spills).
template <typename OutT, int WM, bool AKC, bool BKC, bool FULL>
__global__ __launch_bounds__(256, 3) void bgemm_bf16_kernel(const BGemmArgs p) {
    constexpr int BMT = WM * 32;
    constexpr int ASZ = BMT * HLD > HBK * (BMT + 16) ? BMT * HLD : HBK * (BMT + 16);
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][ASZ + HOPSZ];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    // XCD-aware order: the hardware deals consecutive workgroup ids to the 8 XCDs in turn; remapped so that each XCD (its own
    // L2) walks a CONTIGUOUS range of tiles, with the tiles that share operand rows next to each other: column tile fastest,
    // then row tile, then the tap / head index, then the k split, then the outer batch.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const unsigned L = bx + gx * (by + gy * bz);
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = L & 7, idx = L >> 3;
        const unsigned lg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bx = lg % gx;
        by = (lg / gx) % gy;
        bz = lg / (gx * gy);
    }
    const int b2 = bz % p.nb2;
    const int split = (bz / p.nb2) % splitk;
    const int b1 = bz / (p.nb2 * splitk);
    const int m0 = by * BMT, n0 = bx * BN;

    OperandH A, B;
    A.p = (const unsigned short*)p.A + b1 * p.sA1 + b2 * p.sA2;
    A.s_mn = p.sAm; A.s_k = p.sAk; A.mn0 = m0; A.MN = p.M;
    A.shift_mn = 0; A.shift_k = 0; A.seg = 0; A.vec = p.vecA;
    B.p = (const unsigned short*)p.B + b1 * p.sB1 + b2 * p.sB2;
    B.s_mn = p.sBn; B.s_k = p.sBk; B.mn0 = n0; B.MN = p.N;
    B.shift_mn = 0; B.shift_k = 0; B.seg = 0; B.vec = p.vecB;
    if (p.seg && p.taps <= 1) {
        B.seg = p.seg;
        B.shift_k = p.b_shift0 + b2 * p.b_shift_step;
    }
    constexpr bool akc = AKC, bkc = BKC;
    const int Kin = p.taps > 1 ? p.Kin : p.K;
    const int tiles_per_tap = (Kin + HBK - 1) / HBK;
    const int ntiles = tiles_per_tap * (p.taps > 1 ? p.taps : 1);
    const int per = (ntiles + splitk - 1) / splitk;
    const int t_begin = split * per, t_end = min(ntiles, t_begin + per);

    f32x4_t acc[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned short* Ap0 = A.p;
    const unsigned short* Bp0 = B.p;
    auto fetch = [&](int t, uint4 (&ra)[BMT / 64], uint4 (&rb)[2]) {
        int k0 = t * HBK, kend = p.K;
        if (p.taps > 1) {
            const int tap = t / tiles_per_tap;
            k0 = (t - tap * tiles_per_tap) * HBK;
            kend = Kin;
            A.seg = p.seg;
            A.shift_mn = p.a_shift0 + tap * p.a_shift_step;
            A.p = Ap0;
            B.p = Bp0 + tap * p.sBtap;
        }
        load_tile_h<BMT, AKC>(A, k0, kend, ra);
        load_tile_h<128, BKC>(B, k0, kend, rb);
    };

    auto compute = [&](int cur) {
        const unsigned short* sa = lds[cur];
        const unsigned short* sb = lds[cur] + ASZ;
        uint4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = frag_h(sb, bkc, wn * 64 + j * 16, fr, fg, 128 + 16);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const uint4 a = frag_h(sa, akc, wm * (WM * 16) + i * 16, fr, fg, BMT + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma16<bf16>::step(a, b[j], acc[i][j]);
        }
    };
    uint4 ra[BMT / 64], rb[2];
    const bool plain = FULL || p.taps <= 1;  // wave-uniform: cursors for the plain forms, the generic loader for the implicit-conv dgrad
    TileCursor<BMT, AKC> ca;
    TileCursor<128, BKC> cb;
    if (plain) {
        ca.init(A, t_begin * HBK, p.K);
        cb.init(B, t_begin * HBK, p.K);
    }
    const unsigned short* safeB = (const unsigned short*)p.B;
    if (t_begin < t_end) {
        if constexpr (FULL) { ca.load_full(ra, safeB); cb.load_full(rb, safeB); }
        else if (plain) { ca.load(ra); cb.load(rb); }
        else fetch(t_begin, ra, rb);
        store_tile_h<BMT>(lds[0], akc, ra);
        store_tile_h<128>(lds[0] + ASZ, bkc, rb);
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) {
            if constexpr (FULL) { ca.next(); cb.next(); ca.load_full(ra, safeB); cb.load_full(rb, safeB); }
            else if (plain) { ca.next(); cb.next(); ca.load(ra); cb.load(rb); }
            else fetch(t + 1, ra, rb);
        }
        compute(cur);
        if (more) {
            store_tile_h<BMT>(lds[cur ^ 1], akc, ra);
            store_tile_h<128>(lds[cur ^ 1] + ASZ, bkc, rb);
        }
        __syncthreads();
    }

    if (splitk > 1) {
        float* ws = p.ws + ((long)(b1 * p.nb2 + b2) * splitk + split) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * (WM * 16) + i * 16 + fg * 4 + r, n = n0 + wn * 64 + j * 16 + fr;
                    if (m < p.M && n < p.N) ws[(long)m * p.N + n] = acc[i][j][r];
                }
        return;
    }
    // Epilogue through LDS (free after the last barrier): the MFMA layout gives a lane one column and four rows, i.e. 2- / 4-byte
    // accesses in 32- / 64-byte runs; staged per wave (32 x 64 fp32 at a time, rows padded to 68) a lane gets four consecutive
    // columns of one row, so C, the old C (beta) and the softmax-backward epilogue's P move in 8- / 16-byte pieces, 128 / 256
    // contiguous bytes per row.
    OutT* C = (OutT*)p.C + b1 * p.sC1 + b2 * p.sC2;
    constexpr int SLD = 68;
    static_assert(4 * 32 * SLD * 4 <= 2 * (ASZ + HOPSZ) * 2, "stage does not fit the operand buffers");
    float* stage = (float*)&lds[0][0] + wid * (32 * SLD);
    const bool vec_c = p.ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && (!p.epi_p || ((uintptr_t)p.epi_p & 15) == 0) &&
                       p.sC1 % 4 == 0 && p.sC2 % 4 == 0;
    const long bidx = (long)b1 * p.nb2 + b2;
#pragma unroll
    for (int half = 0; half < WM / 2; ++half) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(i2 * 16 + fg * 4 + r) * SLD + j * 16 + fr] = acc[half * 2 + i2][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + (lane >> 4), cl = (lane & 15) * 4;
            const int m = m0 + wm * (WM * 16) + half * 32 + rl, n = n0 + wn * 64 + cl;
            if (m >= p.M || n >= p.N) continue;
            const float4 a4 = *(const float4*)(stage + rl * SLD + cl);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            OutT* dst = C + (long)m * p.ldc + n;
            const bool full = vec_c && n + 3 < p.N;
            if (p.epi_p) {
                const OutT* pp = (const OutT*)p.epi_p + b1 * p.sC1 + b2 * p.sC2 + (long)m * p.ldc + n;
                float pv[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) ld4<OutT>(pp, pv);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) pv[e] = Num<OutT>::to_f32(pp[e]);
                const float dl = p.epi_delta[bidx * p.M + m];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float dp = v[e];
                    if (p.drop_p > 0.f) {
                        const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
                        dp = dropout_bits(p.drop_seed, p.drop_key, ((uint64_t)bidx * p.M + m) * p.N + n + e) >= thr ? dp * (1.f / (1.f - p.drop_p)) : 0.f;
                    }
                    v[e] = p.alpha * pv[e] * (dp - dl);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = p.alpha * v[e] + ((p.bias && n + e < p.N) ? p.bias[n + e] : 0.f);
            }
            if (p.beta != 0.f) {
                float old[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) ld4<OutT>(dst, old);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) old[e] = Num<OutT>::to_f32(dst[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += p.beta * old[e];
            }
            if (full) st4<OutT>(dst, v);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < p.N) dst[e] = Num<OutT>::from_f32(v[e]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename OutT>
__global__ void bgemm_reduce_t_kernel(BGemmArgs p) {
    const long per = (long)p.M * p.N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int bz = blockIdx.y;
    if (i >= per) return;
    const int b2 = bz % p.nb2, b1 = bz / p.nb2;
    const float* ws = p.ws + (long)bz * p.splitk * per + i;
    float s = 0.f;
    for (int k = 0; k < p.splitk; ++k) s += ws[(long)k * per];
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float v = p.alpha * s + (p.bias ? p.bias[n] : 0.f);
    OutT* dst = (OutT*)p.C + b1 * p.sC1 + b2 * p.sC2 + (long)m * p.ldc + n;
    if (p.beta != 0.f) v += p.beta * Num<OutT>::to_f32(*dst);
    *dst = Num<OutT>::from_f32(v);
}
// ... four columns per thread, four slabs' loads in flight (fp32 C with N, ldc, the batch strides % 4 == 0 and 16-byte bases)
__global__ __launch_bounds__(256) void bgemm_reduce_v4_kernel(BGemmArgs p) {
    const long per4 = (long)p.M * p.N / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int bz = blockIdx.y;
    if (i >= per4) return;
    const int b2 = bz % p.nb2, b1 = bz / p.nb2;
    const float4* ws = (const float4*)(p.ws + (long)bz * p.splitk * per4 * 4) + i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= p.splitk; k += 4) {
        const float4 a = ws[(long)k * per4], b = ws[(long)(k + 1) * per4], c = ws[(long)(k + 2) * per4], d = ws[(long)(k + 3) * per4];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < p.splitk; ++k) {
        const float4 a = ws[(long)k * per4];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const int n4 = p.N / 4, m = (int)(i / n4), n = (int)(i % n4) * 4;
    float4 v = make_float4(p.alpha * s.x, p.alpha * s.y, p.alpha * s.z, p.alpha * s.w);
    if (p.bias) { v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3]; }
    float4* dst = (float4*)((float*)p.C + b1 * p.sC1 + b2 * p.sC2 + (long)m * p.ldc + n);
    if (p.beta != 0.f) {
        const float4 o = *dst;
        v.x += p.beta * o.x; v.y += p.beta * o.y; v.z += p.beta * o.z; v.w += p.beta * o.w;
    }
    *dst = v;
}

// ---- TN products with a long reduction (weight gradients): 256 x 256 x 32 tiles by LDS-DMA --------------------------------
// dW = dY^T X reduces over all B*T rows into a small output, both operands k-strided ("natural": a k row is contiguous
// along m / n).  The 128 x 128 tile above moves 64 flop per operand byte, i.e. it needs ~20 TB/s out of L2 at half the
// MFMA peak; this kernel's 256 x 256 tile needs half of that and, as the forward slab kernel does, keeps its operands off the
// VGPRs: 8 waves (2 x 4, a wave owns 128 x 64 = 8 x 4 accumulator fragments), three 32 KiB LDS stages ([32 k][256 m] of A,
// then of B), `buffer_load ... lds` 16 B per lane (one instruction = two k rows), a counted vmcnt and ONE barrier per
// k-step.  A DMA writes LDS lane-linearly, so the bank swizzle lives on the global side: LDS 32-byte position q of row k
// holds global columns 16 * (q ^ f(k)), f(k) = (k & 3) | ((k >> 3) & 1) << 2, which makes the 32 lanes of a
// ds_read_b64_tr_b16 half (k rows 0-3 and 8-11 of one 16-column block) cover all 64 banks once.  Rows that must read as
// zero (the wgrad form's shifted time rows outside their utterance) carry an out-of-range buffer offset.
constexpr int GT = 256, GK = 32;
constexpr int GOP = GK * GT * 2;   // bytes per operand per stage

__global__ __launch_bounds__(512) void bgemm_tn256_kernel(const BGemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned char st0[2 * GOP];
    __shared__ __attribute__((aligned(16))) unsigned char st1[2 * GOP];
    __shared__ __attribute__((aligned(16))) unsigned char st2[2 * GOP];
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    const int tn = p.N / GT, tm = p.M / GT;
    // XCD-contiguous order (workgroup i runs on XCD i % 8): column tile fastest, then tap, row tile, k split, outer batch - the
    // workgroups that share an A panel (one row tile, one split, every tap / column tile) sit in one XCD's L2 together
    unsigned lg = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = lg & 7, idx = lg >> 3;
        lg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bn = lg % tn; lg /= tn;
    const int b2 = lg % p.nb2; lg /= p.nb2;
    const int bm = lg % tm; lg /= tm;
    const int split = lg % splitk;
    const int b1 = lg / splitk;
    const int m0 = bm * GT, n0 = bn * GT;
    const int ntiles = p.K / GK, per = ntiles / splitk, rem = ntiles % splitk;
    const int t_begin = split * per + (split < rem ? split : rem), nk = per + (split < rem ? 1 : 0);

    const unsigned short* Ab = (const unsigned short*)p.A + b1 * p.sA1 + b2 * p.sA2;
    const unsigned short* Bb = (const unsigned short*)p.B + b1 * p.sB1 + b2 * p.sB2;
    constexpr unsigned OOB = 0xFFFFF000u;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (unsigned)(((long)(p.K - 1) * p.sAk + p.M) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (unsigned)(((long)(p.K - 1) * p.sBk + p.N) * 2), 0x00020000);
    const int shift = p.seg ? p.b_shift0 + b2 * p.b_shift_step : 0;
    const int seg = p.seg ? p.seg : 0x7fffffff;
    const unsigned stepA = (unsigned)(GK * p.sAk * 2), stepB = (unsigned)(GK * p.sBk * 2);
    unsigned voffA[2], voffB[2];
    int rsB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 2 * (wave + 8 * j) + (lane >> 5), pos = lane & 31;
        const int f = (r & 3) | (((r >> 3) & 1) << 2);
        const int c = (((pos >> 1) ^ f) << 1) | (pos & 1);   // global 16-byte chunk of the row that lands at LDS position pos
        voffA[j] = (unsigned)(((long)r * p.sAk + m0 + c * 8) * 2);
        voffB[j] = (unsigned)(((long)(t_begin * GK + r + shift) * p.sBk + n0 + c * 8) * 2);  // wraps when the row is negative: never used then
        rsB[j] = r + shift;
    }
    int t_next = t_begin;  // the k-step the next issue() fetches
    int kq = p.seg ? (t_begin * GK) % p.seg : 0;  // position of the next stage's first k row inside its utterance
    auto issue = [&](unsigned char* dst) {
        const unsigned soA = __builtin_amdgcn_readfirstlane((unsigned)t_next * stepA);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(dst + (wave + 8 * j) * 1024), 16, voffA[j], soA, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned v = (unsigned)(kq + rsB[j]) < (unsigned)seg ? voffB[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)(dst + GOP + (wave + 8 * j) * 1024), 16, v, 0, 0, 0);
            voffB[j] += stepB;
        }
        ++t_next;
        if (p.seg) { kq += GK; if (kq >= p.seg) kq -= p.seg; }
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int fl = (fr >> 2) | ((fg & 1) << 2);
    const int rowoff = (fg * 8 + (fr >> 2)) * (GT * 2) + (fr & 3) * 8;
    int aoff[8], boff[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) aoff[i] = rowoff + (((wm * 8 + i) ^ fl) << 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = GOP + rowoff + (((wn * 4 + j) ^ fl) << 5);
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const unsigned char* st) {
        union Frag { s16x4_t v[2]; bf16x8_t h; };
        Frag fb[4], fa[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fb[j].v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(st + boff[j]));
            fb[j].v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(st + boff[j] + 4 * GT * 2));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[i].v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(st + aoff[i]));
            fa[i].v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(st + aoff[i] + 4 * GT * 2));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i].h, fb[j].h, acc[i][j], 0, 0, 0);
    };
    // stage t has landed once at most the 4 DMAs of stage t+1 are still outstanding; the barrier also frees the stage that
    // step t-1 was multiplied from, which is where stage t+2 goes
#define FS2_TN_STEP(cur, nxt2, tv)                                                \
    {                                                                             \
        if ((tv) + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F74); /* vmcnt(4) */      \
        else __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */      \
        __builtin_amdgcn_s_barrier();                                             \
        if ((tv) + 2 < nk) issue(nxt2);                                           \
        compute(cur);                                                             \
    }
    // steady state with nothing conditional in it (the compiler's own LDS-DMA scoreboard then sees that vmcnt(4) retires the
    // stage about to be read, and adds no vmcnt(0) of its own in front of the fragment reads); the last 2-4 steps take the
    // guarded form
#define FS2_TN_FULL(cur, nxt2)                                                    \
    {                                                                             \
        __builtin_amdgcn_s_waitcnt(0x0F74);                                       \
        __builtin_amdgcn_s_barrier();                                             \
        issue(nxt2);                                                              \
        compute(cur);                                                             \
    }
    int t = 0;
    if (nk >= 5) {
        issue(st0);
        issue(st1);
        for (; t + 5 <= nk; t += 3) {
            FS2_TN_FULL(st0, st2)
            FS2_TN_FULL(st1, st0)
            FS2_TN_FULL(st2, st1)
        }
    } else {
        if (nk > 0) issue(st0);
        if (nk > 1) issue(st1);
    }
    for (; t < nk; t += 3) {
        FS2_TN_STEP(st0, st2, t)
        if (t + 1 < nk) FS2_TN_STEP(st1, st0, t + 1)
        if (t + 2 < nk) FS2_TN_STEP(st2, st1, t + 2)
    }
#undef FS2_TN_FULL
#undef FS2_TN_STEP

    if (splitk > 1) {
        float* ws = p.ws + ((long)(b1 * p.nb2 + b2) * splitk + split) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 128 + i * 16 + fg * 4 + r, n = n0 + wn * 64 + j * 16 + fr;
                    ws[(long)m * p.N + n] = acc[i][j][r];
                }
        return;
    }
    float* C = (float*)p.C + b1 * p.sC1 + b2 * p.sC2;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 128 + i * 16 + fg * 4 + r, n = n0 + wn * 64 + j * 16 + fr;
                float* dst = C + (long)m * p.ldc + n;
                float v = p.alpha * acc[i][j][r] + (p.bias ? p.bias[n] : 0.f);
                if (p.beta != 0.f) v += p.beta * *dst;
                *dst = v;
            }
#endif
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// A/B knobs (Tuning): bgemm_full - the bounds-free instantiation for full, aligned tiles; bgemm_xcd - XCD-contiguous tile order of
// the bf16 kernel; bgemm_tn256 - the 256 x 256 LDS-DMA kernel for eligible TN products (fs2_op_bgemm_tn256)

// bf16 TN product in the plain or wgrad form, whole 256 x 256 x 32 tiles, fp32 output, operands inside 32-bit buffer offsets
bool bgemm_tn256_eligible(const BGemmArgs& a) {
    if (!tuning_of(a.tune).bgemm_tn256 || a.sAm != 1 || a.sBn != 1 || a.sAk == 1 || a.sBk == 1) return false;
    if (a.M % GT || a.N % GT || a.K % GK || a.taps > 1 || a.c_dtype != FS2_F32 || a.epi_p) return false;
    if (a.seg && (a.seg % GK != 0 || a.K % a.seg != 0)) return false;  // (seg == 0: the k shifts are not applied, as in the general kernel)
    if (!aligned16(a.A) || !aligned16(a.B) || a.sAk % 8 || a.sBk % 8 || a.sA1 % 8 || a.sA2 % 8 || a.sB1 % 8 || a.sB2 % 8) return false;
    const long abytes = ((long)(a.K - 1) * a.sAk + a.M) * 2, bbytes = ((long)(a.K - 1) * a.sBk + a.N) * 2;
    return abytes < 0xFFFFF000L && bbytes < 0xFFFFF000L;
}

size_t bgemm_ws_bytes(const BGemmArgs& a) {
    return a.splitk > 1 ? (size_t)a.nb1 * a.nb2 * a.splitk * a.M * a.N * sizeof(float) : 0;
}

// the split-K reduce pass of the bf16 kernels into an fp32 C: the 16-byte form when the layout allows it (same sums, same order)
static void launch_reduce_f32(const BGemmArgs& a, dim3 g2, hipStream_t stream) {
    const bool v4 = a.N % 4 == 0 && a.ldc % 4 == 0 && a.sC1 % 4 == 0 && a.sC2 % 4 == 0 && aligned16(a.C) && aligned16(a.ws) &&
                    (!a.bias || aligned16(a.bias));
    if (v4) {
        const long per4 = (long)a.M * a.N / 4;
        hipLaunchKernelGGL(bgemm_reduce_v4_kernel, dim3((unsigned)((per4 + 255) / 256), g2.y), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(bgemm_reduce_t_kernel<float>, g2, dim3(256), 0, stream, a);
    }
}

int launch_bgemm(const BGemmArgs& a0, int dtype, hipStream_t stream) {
    if (dtype != FS2_F32 && dtype != FS2_BF16) return FS2_ERR_SHAPE;
    BGemmArgs a = a0;
    if (dtype == FS2_F32 && a.c_dtype != FS2_F32) return FS2_ERR_SHAPE;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.nb1 <= 0 || a.nb2 <= 0) return FS2_ERR_SHAPE;
    if ((a.sAm != 1 && a.sAk != 1) || (a.sBk != 1 && a.sBn != 1)) return FS2_ERR_SHAPE;
    if (a.taps > 1 && (a.Kin <= 0 || a.K != a.taps * a.Kin)) return FS2_ERR_SHAPE;
    if (a.splitk > 1 && !a.ws) return FS2_ERR_ARG;
    if (a.epi_p && (a.splitk > 1 || !a.epi_delta || a.bias)) return FS2_ERR_ARG;
    // 16-byte vector loads need every row start (and batch / tap base) on a 16-byte boundary
    const int per16 = dtype == FS2_F32 ? 4 : 8;
    auto vec_ok = [per16](const void* p, long s_outer, long s1, long s2, long s3) {
        return aligned16(p) && s_outer % per16 == 0 && s1 % per16 == 0 && s2 % per16 == 0 && s3 % per16 == 0;
    };
    a.vecA = vec_ok(a.A, a.sAk == 1 ? a.sAm : a.sAk, a.sA1, a.sA2, 0) ? 1 : 0;
    a.vecB = vec_ok(a.B, a.sBk == 1 ? a.sBn : a.sBk, a.sB1, a.sB2, a.taps > 1 ? a.sBtap : 0) ? 1 : 0;
    a.xcd_remap = tuning_of(a.tune).bgemm_xcd;
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.nb1 * a.nb2 * splitk);
    const long per = (long)a.M * a.N;
    dim3 g2((unsigned)((per + 255) / 256), a.nb1 * a.nb2);
    if (dtype == FS2_F32) {
        const bool akc = a.sAk == 1, bkc = a.sBk == 1;
        if (akc && bkc) hipLaunchKernelGGL((bgemm_f32_kernel<true, true>), grid, dim3(256), 0, stream, a);
        else if (akc) hipLaunchKernelGGL((bgemm_f32_kernel<true, false>), grid, dim3(256), 0, stream, a);
        else if (bkc) hipLaunchKernelGGL((bgemm_f32_kernel<false, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((bgemm_f32_kernel<false, false>), grid, dim3(256), 0, stream, a);
        if (splitk > 1) hipLaunchKernelGGL(bgemm_reduce_kernel, g2, dim3(256), 0, stream, a);
    } else {
        const bool akc = a.sAk == 1, bkc = a.sBk == 1;
        if (bgemm_tn256_eligible(a)) {
            const unsigned nwg = (unsigned)((a.M / GT) * (a.N / GT) * a.nb1 * a.nb2 * splitk);
            hipLaunchKernelGGL(bgemm_tn256_kernel, dim3(nwg), dim3(512), 0, stream, a);
            if (splitk > 1) launch_reduce_f32(a, g2, stream);
            return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
        }
        // (measured: a win for the TN products - weight gradients, dV, dK: conv1 wgrad 377 -> 350 us, conv2 wgrad 86 -> 67 - and a
        // loss with a k-contiguous A - P V 87 -> 136 us - where the fourth wave per SIMD only adds LDS pressure: TN only)
        const bool full = tuning_of(a.tune).bgemm_full && !akc && a.M % BM == 0 && a.N % BN == 0 && a.K % HBK == 0 && a.vecA && a.vecB && a.taps <= 1 &&
                          (a.seg == 0 || (a.seg >= HBK && !bkc));
#define FS2_BG2(OT, FL) \
        do { \
            if (akc && bkc) hipLaunchKernelGGL((bgemm_bf16_kernel<OT, 4, true, true, FL>), grid, dim3(256), 0, stream, a); \
            else if (akc) hipLaunchKernelGGL((bgemm_bf16_kernel<OT, 4, true, false, FL>), grid, dim3(256), 0, stream, a); \
            else if (bkc) hipLaunchKernelGGL((bgemm_bf16_kernel<OT, 4, false, true, FL>), grid, dim3(256), 0, stream, a); \
            else hipLaunchKernelGGL((bgemm_bf16_kernel<OT, 4, false, false, FL>), grid, dim3(256), 0, stream, a); \
        } while (0)
#define FS2_BG(OT) do { if (full) FS2_BG2(OT, true); else FS2_BG2(OT, false); } while (0)
        if (a.c_dtype == FS2_F32) {
            FS2_BG(float);
            if (splitk > 1) launch_reduce_f32(a, g2, stream);
        } else {
            FS2_BG(bf16);
            if (splitk > 1) hipLaunchKernelGGL(bgemm_reduce_t_kernel<bf16>, g2, dim3(256), 0, stream, a);
        }
#undef FS2_BG
#undef FS2_BG2
    }
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
