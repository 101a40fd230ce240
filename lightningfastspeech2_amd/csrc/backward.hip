// Row / column kernels of the training step's backward (the reference gets them from autograd,
// fastspeech2.py:786-797): LayerNorm backward, column sums (bias / gamma / beta / speaker gradients), masked softmax
// forward + backward for the materialised attention of the training path, ReLU mask, embedding scatter, length-regulator
// segment sums, masked-mean loss gradients, global gradient norm and the AdamW update (fastspeech2.py:1166-1173).
// All HBM-bound: 16-byte accesses, one wave per row for the row kernels, fixed-order reductions (bit-equal reruns).
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

namespace {

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

// ---- LayerNorm backward -------------------------------------------------------------------------------------------
// y = (z - mean) * rstd * gamma + beta.  dz = rstd * (g - mean(g) - zhat * mean(g * zhat)), g = dy * gamma.
// One wave per row, 4 waves per workgroup, LN_ROWS rows per workgroup, a lane owns 4 consecutive columns of every
// 256-column chunk (8- / 16-byte accesses); the column sums of dy * zhat, dy and (optionally ReLU-masked) dz are kept in
// registers over the workgroup's rows and leave as one partial per workgroup: (dgamma, dbeta, dbias of the layer below).
constexpr int LN_ROWS = 32;
constexpr int LN_MAXC = 4;  // 256-column chunks: H <= 1024

template <typename T, int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(LayerNormBwdArgs p) {
    __shared__ float red[4][3][256 * NC];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const T* z = (const T*)p.z;
    const T* zr = (const T*)p.res;
    const T* dy = (const T*)p.dy;
    T* dz = (T*)p.dz;
    const int H = p.H;
    float dg[NC][4], db[NC][4], dc[NC][4], gam[NC][4];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dg[j][e] = db[j][e] = dc[j][e] = 0.f;
            const int c = j * 256 + lane * 4 + e;
            gam[j][e] = c < H ? p.gamma[c] : 0.f;
        }
    const int row0 = blockIdx.x * LN_ROWS;
    const float invH = 1.f / H;
    for (int rr = wid; rr < LN_ROWS; rr += 4) {
        const int row = row0 + rr;
        if (row >= p.M) break;
        float zv[NC][4], dv[NC][4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < H) {
                ld4<T>(z + (long)row * H + c, zv[j]);
                ld4<T>(dy + (long)row * H + c, dv[j]);
                if (p.drop_p > 0.f) {  // dy arrives as the gradient of dropout(y): the forward's mask, regenerated
                    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
                    const float sc = 1.f / (1.f - p.drop_p);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[j][e] = dropout_bits(p.drop_seed, p.drop_key, (uint64_t)row * H + c + e) >= thr ? dv[j][e] * sc : 0.f;
                }
                if (zr) {
                    float r[4];
                    ld4<T>(zr + (long)row * H + c, r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) zv[j][e] += r[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) zv[j][e] = dv[j][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) s += zv[j][e];
        }
        const float mean = wave_sum(s) * invH;
        float q = 0.f;
        bool pos[NC][4];
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pos[j][e] = zv[j][e] > 0.f;  // z is a ReLU output when relu_mask is set
                const float d = (j * 256 + lane * 4 + e < H) ? zv[j][e] - mean : 0.f;
                zv[j][e] = d;
                q += d * d;
            }
        const float rstd = rsqrtf(wave_sum(q) * invH + p.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zv[j][e] *= rstd;  // zhat
                const float g = dv[j][e] * gam[j][e];
                s1 += g;
                s2 += g * zv[j][e];
                dg[j][e] += dv[j][e] * zv[j][e];
                db[j][e] += dv[j][e];
            }
        s1 = wave_sum(s1) * invH;
        s2 = wave_sum(s2) * invH;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = rstd * (dv[j][e] * gam[j][e] - s1 - zv[j][e] * s2);
                if (p.relu_mask && !pos[j][e]) v = 0.f;
                v = Num<T>::to_f32(Num<T>::from_f32(v));  // the column sum is of the stored value
                o[e] = v;
                if (!p.dzm) dc[j][e] += v;
            }
            const int c = j * 256 + lane * 4;
            if (c < H) st4<T>(dz + (long)row * H + c, o);
            if (p.dzm) {  // the same gradient through the dropout on the sub-layer's summand: a second tensor instead of a pass of its own
                const uint32_t thr = (uint32_t)(p.out_p * 16777216.0f);
                const float sc = 1.f / (1.f - p.out_p);
                float m[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = dropout_bits(p.drop_seed, p.out_key, (uint64_t)row * H + c + e) >= thr ? o[e] * sc : 0.f;
                    v = Num<T>::to_f32(Num<T>::from_f32(v));
                    m[e] = v;
                    dc[j][e] += v;  // the third column sum is that of dzm
                }
                if (c < H) st4<T>((T*)p.dzm + (long)row * H + c, m);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wid][0][j * 256 + lane * 4 + e] = dg[j][e];
            red[wid][1][j * 256 + lane * 4 + e] = db[j][e];
            red[wid][2][j * 256 + lane * 4 + e] = dc[j][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            p.part[((long)blockIdx.x * 3 + k) * H + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
}

// ---- column sums ----------------------------------------------------------------------------------------------------
// out[s][n] (+)= scale * sum_{rows of segment s} x[row][n].  Pass 1: a workgroup owns 64 columns x one chunk of CS_CHUNK rows
// of one segment (4 row groups x 64 lanes, 8 loads in flight per thread, fixed order) and leaves its partial in ws; pass 2:
// one workgroup per (64 columns, segment) adds the chunk partials the same way.  A one-launch form exists behind a knob
// (g_colsum_fused: the workgroup that takes the last ticket of its (64 columns, segment) counter does pass 2's work, in chunk
// order, and returns the counter to zero; the counters are the first CS_CTR words of ws) - measured slower, see the knob.  Columns n >= n1 may go to a second destination (LayerNorm backward: dgamma|dbeta and the
// producing layer's bias gradient out of one (parts, 3H) array).
constexpr int CS_CHUNK = 128;
constexpr int CS_CTR = 8192;

template <typename T>
__device__ inline float cs_accumulate(const T* x, long ldx, int c, int r0, int r1, int g, const float* w = nullptr) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int r = r0 + g;
    if (w) {  // rows weighted: sum_r w[r] x[r][c] (a vector-matrix product whose vector is a loss gradient)
        for (; r + 28 < r1; r += 32) {
            float xv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = Num<T>::to_f32(x[(long)(r + 4 * u) * ldx + c]); wv[u] = w[r + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = fmaf(xv[u], wv[u], a[u]);
        }
        for (; r < r1; r += 4) a[0] = fmaf(Num<T>::to_f32(x[(long)r * ldx + c]), w[r], a[0]);
    } else {
        for (; r + 28 < r1; r += 32)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += Num<T>::to_f32(x[(long)(r + 4 * u) * ldx + c]);
        for (; r < r1; r += 4) a[0] += Num<T>::to_f32(x[(long)r * ldx + c]);
    }
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

__device__ inline void cs_store(const ColSumArgs& p, int s, int c, float a) {
    a *= p.scale;
    if (p.out2 && c >= p.n1) {
        float* o = p.out2 + (long)s * (p.N - p.n1) + (c - p.n1);
        *o = p.accumulate2 ? *o + a : a;
    } else {
        float* o = p.out + (long)s * (p.out2 ? p.n1 : p.N) + c;
        *o = p.accumulate ? *o + a : a;
    }
}

// FUSED: the last workgroup of a (column block, segment) finishes the sum; otherwise col_sum_pass2 does
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void col_sum_pass1(ColSumArgs p, int nchunk) {
    __shared__ float red[4][64];
    __shared__ int last;
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, g = threadIdx.x >> 6;
    const int chunk = blockIdx.y, s = blockIdx.z;
    const int seg = p.seg > 0 ? p.seg : p.M;
    const int r0 = chunk * CS_CHUNK, r1 = min(seg, r0 + CS_CHUNK);
    float* part = p.ws + CS_CTR;
    red[g][l] = c < p.N ? cs_accumulate<T>((const T*)p.x + (long)s * seg * p.ldx, p.ldx, c, r0, r1, g, p.row_w ? p.row_w + (long)s * seg : nullptr) : 0.f;
    __syncthreads();
    if (g == 0 && c < p.N) {
        const float a = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        if (nchunk == 1) cs_store(p, s, c, a);
        else part[((long)s * nchunk + chunk) * p.N + c] = a;
    }
    if (!FUSED || nchunk == 1) return;
    __threadfence();   // this workgroup's partial is visible device-wide before its ticket is
    __syncthreads();
    unsigned* ctr = (unsigned*)p.ws + (s * gridDim.x + blockIdx.x);
    if (threadIdx.x == 0) last = atomicAdd(ctr, 1u) == (unsigned)(nchunk - 1);
    __syncthreads();
    if (!last) return;
    __threadfence();
    red[g][l] = c < p.N ? cs_accumulate<float>(part + (long)s * nchunk * p.N, p.N, c, 0, nchunk, g) : 0.f;
    __syncthreads();
    if (g == 0 && c < p.N) cs_store(p, s, c, (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]));
    if (threadIdx.x == 0) *ctr = 0u;
}
__global__ __launch_bounds__(256) void col_sum_pass2(ColSumArgs p, int nchunk) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, g = threadIdx.x >> 6, s = blockIdx.y;
    red[g][l] = c < p.N ? cs_accumulate<float>(p.ws + CS_CTR + (long)s * nchunk * p.N, p.N, c, 0, nchunk, g) : 0.f;
    __syncthreads();
    if (g == 0 && c < p.N) cs_store(p, s, c, (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]));
}

// ---- masked softmax over the key axis (training path: probabilities are materialised; HBM is 288 GB) --------------
// one wave per (b, head, query) row.  Scores and dP are fp32 (the GEMM that makes them writes fp32 in either precision
// mode); probabilities and dS are in the activation dtype T (they are MFMA operands next).  p / out may alias s for fp32.
// single pass: the row (S <= 256 * NV floats, S % 4 == 0) stays in registers, 16-byte loads
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_fwd_row_kernel(SoftmaxArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.B * p.heads * p.S;
    if (row >= rows) return;
    const int b = (int)(row / ((long)p.heads * p.S));
    const float* s = p.s + row * p.S;
    T* out = (T*)p.out + row * p.S;
    const uint8_t* pad = p.key_pad ? p.key_pad + (long)b * p.S : nullptr;
    float v[NV][4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k = j * 256 + lane * 4;
        if (k < p.S) {
            ld4<float>(s + k, v[j]);
            const uint32_t pm = pad ? *(const uint32_t*)(pad + k) : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] = ((pm >> (8 * e)) & 0xffu) ? -INFINITY : v[j][e] * p.scale;
                mx = fmaxf(mx, v[j][e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = -INFINITY;
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = v[j][e] == -INFINITY ? 0.f : expf(v[j][e] - mx);
            sum += v[j][e];
        }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k = j * 256 + lane * 4;
        if (k < p.S) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] *= inv;
            st4<T>(out + k, v[j]);
        }
    }
}
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_bwd_row_kernel(SoftmaxArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.B * p.heads * p.S;
    if (row >= rows) return;
    const float* d = p.s + row * p.S;
    const T* pr = (const T*)p.p + row * p.S;
    T* out = (T*)p.out + row * p.S;
    float dv[NV][4], pv[NV][4];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k = j * 256 + lane * 4;
        if (k < p.S) {
            ld4<float>(d + k, dv[j]);
            ld4<T>(pr + k, pv[j]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[j][e] = pv[j][e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += dv[j][e] * pv[j][e];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k = j * 256 + lane * 4;
        if (k < p.S) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[j][e] = p.scale * pv[j][e] * (dv[j][e] - dot);
            st4<T>(out + k, dv[j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(SoftmaxArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.B * p.heads * p.S;
    if (row >= rows) return;
    const int b = (int)(row / ((long)p.heads * p.S));
    const float* s = (const float*)p.s + row * p.S;
    T* out = (T*)p.out + row * p.S;
    const uint8_t* pad = p.key_pad ? p.key_pad + (long)b * p.S : nullptr;
    float mx = -INFINITY;
    for (int k = lane; k < p.S; k += 64) {
        const float v = (pad && pad[k]) ? -INFINITY : s[k] * p.scale;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < p.S; k += 64) sum += (pad && pad[k]) ? 0.f : expf(s[k] * p.scale - mx);
    const float inv = 1.f / wave_sum(sum);
    for (int k = lane; k < p.S; k += 64) out[k] = Num<T>::from_f32((pad && pad[k]) ? 0.f : expf(s[k] * p.scale - mx) * inv);
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(SoftmaxArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.B * p.heads * p.S;
    if (row >= rows) return;
    const float* d = (const float*)p.s + row * p.S;
    const T* pr = (const T*)p.p + row * p.S;
    T* out = (T*)p.out + row * p.S;
    float dot = 0.f;
    for (int k = lane; k < p.S; k += 64) dot += d[k] * Num<T>::to_f32(pr[k]);
    dot = wave_sum(dot);
    for (int k = lane; k < p.S; k += 64) out[k] = Num<T>::from_f32(p.scale * Num<T>::to_f32(pr[k]) * (d[k] - dot));
}

// 16 bytes per thread and operand when everything is 16-byte aligned (the activation tensors are); scalar tail / fallback
template <typename T>
__global__ void ew_vec_kernel(EwArgs p) {
    constexpr int NE = Vec16<T>::N;
    const size_t nv = p.n / NE;
    const uint4* a = (const uint4*)p.a;
    const uint4* b = (const uint4*)p.b;
    uint4* out = (uint4*)p.out;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nv; e += (size_t)gridDim.x * blockDim.x) {
        float fa[NE], fb[NE];
        Vec16<T>::unpack(a[e], fa);
        if (b) Vec16<T>::unpack(b[e], fb);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (p.op == 0) fa[i] = p.alpha * fa[i] + (b ? p.beta * fb[i] : 0.f);
            else if (p.op == 1) fa[i] = fb[i] > 0.f ? fa[i] : 0.f;
            else fa[i] = p.alpha * fa[i];
        }
        out[e] = Vec16<T>::pack(fa);
    }
}

template <typename T>
__global__ void ew_kernel(EwArgs p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const T* a = (const T*)p.a;
    const T* b = (const T*)p.b;
    T* out = (T*)p.out;
    for (size_t e = i; e < p.n; e += stride) {
        float v;
        if (p.op == 0) v = p.alpha * Num<T>::to_f32(a[e]) + (b ? p.beta * Num<T>::to_f32(b[e]) : 0.f);
        else if (p.op == 1) v = Num<T>::to_f32(b[e]) > 0.f ? Num<T>::to_f32(a[e]) : 0.f;
        else v = p.alpha * Num<T>::to_f32(a[e]);
        out[e] = Num<T>::from_f32(v);
    }
}

// ---- dropout: counter-based mask (splitmix64 of seed, site key, element index), regenerated - never stored -------------
// y = x * keep / (1 - p); the same (seed, key) gives the same mask, so the backward applies the same launch to the gradient.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(DropoutArgs p) {
    const uint32_t thr = (uint32_t)(p.p * 16777216.0f);  // drop when bits < thr
    const float scale = 1.f / (1.f - p.p);
    const T* x = (const T*)p.x;
    T* y = (T*)p.y;
    constexpr int V = 16 / (int)sizeof(T);  // 16-byte accesses (the first version moved 2 bytes per thread: 73 us for 100 MB)
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const size_t nv = vec ? p.n / V : 0;
    for (size_t q = tid; q < nv; q += stride) {
        float f[V];
        Vec16<T>::unpack(((const uint4*)x)[q], f);
#pragma unroll
        for (int e = 0; e < V; ++e) f[e] = dropout_bits(p.seed, p.key, q * V + e) >= thr ? f[e] * scale : 0.f;
        ((uint4*)y)[q] = Vec16<T>::pack(f);
    }
    for (size_t e = nv * V + tid; e < p.n; e += stride) {
        const bool keep = dropout_bits(p.seed, p.key, e) >= thr;
        y[e] = Num<T>::from_f32(keep ? Num<T>::to_f32(x[e]) * scale : 0.f);
    }
}

// delta[b][h][q] = sum_d dO[b*S + q][h*d .. ] * O[...]  (= sum_k dP P of the attention backward); one wave per (row, head)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnDeltaArgs p) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.B * p.S;
    if (w >= rows * p.heads) return;
    const long row = w / p.heads;
    const int h = (int)(w % p.heads), d = p.H / p.heads;
    const T* a = (const T*)p.dout + row * p.H + h * d;
    const T* o = (const T*)p.out + row * p.H + h * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s = fmaf(Num<T>::to_f32(a[c]), Num<T>::to_f32(o[c]), s);
    s = wave_sum(s);
    if (lane == 0) p.delta[((row / p.S) * p.heads + h) * p.S + row % p.S] = s;
}

// pred[m] = mask[m] ? 0 : y[m] . w + b   (the predictor head after a dropout layer; one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void row_dot_kernel(RowDotArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const T* y = (const T*)p.y + row * p.H;
    float a = 0.f;
    for (int c = lane; c < p.H; c += 64) a = fmaf(Num<T>::to_f32(y[c]), p.w[c], a);
    a = wave_sum(a);
    if (lane == 0) p.pred[row] = (p.mask && p.mask[row]) ? 0.f : a + p.b[0];
}

// ---- embedding backward: one workgroup per table row, source rows visited in index order --------------------------
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(ScatterRowsArgs p) {
    const int v = blockIdx.x;
    if (v == p.skip_row) return;
    // Blocks of 4096 source rows: wave w scans its own 1024 of them (64 at a time, ballot + popcount compaction into its own
    // list, no workgroup barrier), then all threads add the listed rows wave list by wave list - ascending row order, so the
    // sums do not depend on scheduling.
    __shared__ int hits[4][1024];
    __shared__ int cnt[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c0 = 0; c0 < p.H; c0 += 256 * 8) {  // register accumulators for up to 8 columns per thread per sweep
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int r0 = 0; r0 < p.R; r0 += 4096) {
            int n = 0;
            int iv[16];  // all 16 index loads in flight before the first is looked at
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = r0 + w * 1024 + q * 64 + lane;
                iv[q] = r < p.R ? (p.idx32 ? p.idx32[r] : (int)p.idx64[r]) : -1;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = r0 + w * 1024 + q * 64 + lane;
                const bool m = iv[q] == v;
                const unsigned long long bal = __ballot(m);
                if (m) hits[w][n + __popcll(bal & ((1ull << lane) - 1))] = r;
                n += __popcll(bal);
            }
            if (lane == 0) cnt[w] = n;
            __syncthreads();
#pragma unroll 1
            for (int ww = 0; ww < 4; ++ww) {
                const int nh = cnt[ww];
                // eight rows' loads in flight per column group, added in list order
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + j * 256 + threadIdx.x;
                    if (c0 + j * 256 >= p.H) break;  // uniform
                    for (int h = 0; h < nh; h += 8) {
                        float vv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            vv[u] = (h + u < nh && c < p.H) ? Num<T>::to_f32(((const T*)p.x)[(long)hits[ww][h + u] * p.H + c]) : 0.f;
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[j] += vv[u];
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j * 256 + threadIdx.x;
            if (c < p.H) p.table[(long)v * p.H + c] += acc[j];
        }
    }
}

// Chunked form for long index lists over small tables (the variance embeddings: 49152 frames into 256 buckets): the
// one-workgroup-per-table-row kernel above makes every workgroup scan ALL R indices and leaves 256 workgroups to gather
// 25 MB (132-170 us).  Here a workgroup owns SR_CHUNK source rows x 64 columns and a V x 64 fp32 table in LDS; wave w lists,
// in ascending order, the chunk's rows whose table row v has v % 4 == w (so every LDS cell has ONE writer, which adds its
// rows in order: deterministic), gathers them 16 loads at a time and dumps the table as a partial; the partials are then
// column-summed over the chunks into the table (launch_col_sum, fixed order).
constexpr int SR_CHUNK = 512, SR_VMAX = 256;
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_part_kernel(ScatterRowsArgs p) {
    __shared__ float tab[SR_VMAX * 64];
    __shared__ unsigned hits[4][SR_CHUNK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = blockIdx.x, c = blockIdx.y * 64 + lane;
    const int r0 = chunk * SR_CHUNK;
    for (int i = threadIdx.x; i < p.V * 64; i += 256) tab[i] = 0.f;
    int iv[SR_CHUNK / 64];
#pragma unroll
    for (int q = 0; q < SR_CHUNK / 64; ++q) {
        const int r = r0 + q * 64 + lane;
        iv[q] = r < p.R ? (p.idx32 ? p.idx32[r] : (int)p.idx64[r]) : -1;
    }
    int n = 0;
#pragma unroll
    for (int q = 0; q < SR_CHUNK / 64; ++q) {
        const bool m = iv[q] >= 0 && iv[q] < p.V && iv[q] != p.skip_row && (iv[q] & 3) == w;
        const unsigned long long bal = __ballot(m);
        if (m) hits[w][n + __popcll(bal & ((1ull << lane) - 1))] = ((unsigned)iv[q] << 16) | (unsigned)(q * 64 + lane);
        n += __popcll(bal);
    }
    __syncthreads();
    const T* x = (const T*)p.x;
    for (int h = 0; h < n; h += 16) {
        float vv[16];
        unsigned hv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            hv[u] = hits[w][min(h + u, n - 1)];
            vv[u] = (h + u < n && c < p.H) ? Num<T>::to_f32(x[(long)(r0 + (int)(hv[u] & 0xffffu)) * p.H + c]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (h + u < n) tab[(hv[u] >> 16) * 64 + lane] += vv[u];
    }
    __syncthreads();
    float* part = p.ws + (long)chunk * p.V * p.H;
    if (c < p.H)
        for (int v = w; v < p.V; v += 4) part[(long)v * p.H + c] = tab[v * 64 + lane];
}

// ---- length regulator backward: one wave per (b, phone) -----------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void regulate_bwd_kernel(RegulateBwdArgs p) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (long)p.B * p.L) return;
    const int b = (int)(idx / p.L), ph = (int)(idx % p.L);
    const int t1 = min(p.cum[idx], p.T), t0 = min(ph ? p.cum[idx - 1] : 0, p.T);
    for (int c = lane; c < p.H; c += 64) {
        float a = 0.f;
        for (int t = t0; t < t1; ++t) a += Num<T>::to_f32(((const T*)p.dy)[((long)b * p.T + t) * p.H + c]);
        ((T*)p.dx)[idx * p.H + c] = Num<T>::from_f32(a);
    }
}

__global__ void masked_loss_bwd_kernel(LossBwdArgs p) {
    const int64_t total = p.rows * (int64_t)p.inner;
    const float cnt = p.stat[1];
    const float k = p.alpha / cnt;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / p.inner;
        float g = 0.f;
        if (!p.mask[r]) {
            float t;
            if (p.truth_kind == 0) t = ((const float*)p.truth)[e];
            else t = logf((float)((const int64_t*)p.truth)[e] + 1.0f);
            const float d = p.pred[e] - t;
            g = p.kind == 0 ? (d > 0.f ? k : (d < 0.f ? -k : 0.f)) : 2.f * d * k;
        }
        p.dpred[e] = g;
    }
}

// ---- global gradient norm + AdamW ------------------------------------------------------------------------------------
constexpr int SS_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sum_sq_pass1(const float* x, size_t n, float* ws) {
    __shared__ double red[256];
    // 16-byte loads, fp32 partial sums of at most 64 squares, fp64 across them (x is 16-byte aligned: a flat buffer)
    double a = 0.0;
    const size_t nv = ((uintptr_t)x & 15) == 0 ? n / 4 : 0;
    const float4* xv = (const float4*)x;
    float f = 0.f;
    int run = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nv; e += (size_t)gridDim.x * 256) {
        const float4 v = xv[e];
        f += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        if (++run == 16) { a += (double)f; f = 0.f; run = 0; }
    }
    a += (double)f;
    for (size_t e = nv * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) a += (double)x[e] * x[e];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ((double*)ws)[blockIdx.x] = red[0];
}
__global__ void sum_sq_pass2(const float* ws, int nb, float* out) {
    if (threadIdx.x || blockIdx.x) return;
    double a = 0.0;
    for (int b = 0; b < nb; ++b) a += ((const double*)ws)[b];
    *out = (float)a;
}

// torch.optim.AdamW (fastspeech2.py:1166-1173): decoupled weight decay, bias-corrected moments; the gradient is first
// scaled by grad_scale and by the global-norm clip coefficient min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_).
__device__ inline float adamw_one(const AdamWArgs& p, float& w, float g, float& m, float& v, float gs, float step_size, float inv_sqrt_bc2) {
    g *= gs;
    w *= 1.f - p.lr * p.weight_decay;
    m = p.beta1 * m + (1.f - p.beta1) * g;
    v = p.beta2 * v + (1.f - p.beta2) * g * g;
    w -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + p.eps);
    return w;
}
// Four elements per thread per pass (16-byte accesses; the buffers are torch allocations, n's tail goes element by element);
// with `shadow` the new weights are also written in bf16 - the mixed-precision path's MFMA operands - instead of a separate
// convert pass re-reading all of p.
__global__ __launch_bounds__(256) void adamw_kernel(AdamWArgs p) {
    float gs = p.grad_scale;
    if (p.gnorm_sq) {
        const float norm = sqrtf(*p.gnorm_sq) * p.grad_scale;
        const float coef = p.max_norm / (norm + 1e-6f);
        if (coef < 1.f) gs *= coef;
    }
    const float bc1 = 1.f - powf(p.beta1, (float)p.step), bc2 = 1.f - powf(p.beta2, (float)p.step);
    const float step_size = p.lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const size_t n4 = p.n / 4, stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bf16* sh = (bf16*)p.shadow;
    for (size_t q = tid; q < n4; q += stride) {
        float4 w = ((float4*)p.p)[q], m = ((float4*)p.m)[q], v = ((float4*)p.v)[q];
        const float4 g = ((const float4*)p.g)[q];
        adamw_one(p, w.x, g.x, m.x, v.x, gs, step_size, inv_sqrt_bc2);
        adamw_one(p, w.y, g.y, m.y, v.y, gs, step_size, inv_sqrt_bc2);
        adamw_one(p, w.z, g.z, m.z, v.z, gs, step_size, inv_sqrt_bc2);
        adamw_one(p, w.w, g.w, m.w, v.w, gs, step_size, inv_sqrt_bc2);
        ((float4*)p.p)[q] = w;
        ((float4*)p.m)[q] = m;
        ((float4*)p.v)[q] = v;
        if (sh) {
            const float o[4] = {w.x, w.y, w.z, w.w};
            *(uint2*)(sh + 4 * q) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
    }
    for (size_t e = 4 * n4 + tid; e < p.n; e += stride) {
        float w = p.p[e], m = p.m[e], v = p.v[e];
        adamw_one(p, w, p.g[e], m, v, gs, step_size, inv_sqrt_bc2);
        p.p[e] = w;
        p.m[e] = m;
        p.v[e] = v;
        if (sh) sh[e] = Num<bf16>::from_f32(w);
    }
}

// ---- weights for the data-gradient convolution: dst[ci][j' * N + n] = src[n][(taps - 1 - j') * Cin + ci] -------------
// (the transposed, tap-flipped kernel: dX = conv(dY, dst) with the forward GEMM / slab-conv kernels).  32 x 32 LDS tiles.
template <typename T>
__global__ __launch_bounds__(256) void transpose_weight_kernel(TransposeWeightArgs p) {
    __shared__ T tile[32][33];
    const int tap = blockIdx.z, n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const T* src = (const T*)p.src + (long)tap * p.Cin;
    T* dst = (T*)p.dst + (long)(p.taps - 1 - tap) * p.N;
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < p.N && c0 + tx < p.Cin) tile[r][tx] = src[(long)(n0 + r) * p.taps * p.Cin + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < p.Cin && n0 + tx < p.N) dst[(long)(c0 + r) * p.taps * p.N + n0 + tx] = tile[tx][r];
}

// All of a model's data-gradient weights in ONE launch (the per-weight launches above are 72 per optimizer step at C2, 157 at
// C5): tab = n rows of {src, dst, N, Cin, taps, first tile}, 64 x 64 tiles moved as bf16 pairs.
__global__ __launch_bounds__(256) void transpose_weight_batch_kernel(const long long* __restrict__ tab, int n) {
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = n - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {  // last row whose first tile is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid * 6 + 5] <= b) lo = mid; else hi = mid - 1;
    }
    const long long* d = tab + lo * 6;
    const unsigned short* src = (const unsigned short*)d[0];
    unsigned short* dst = (unsigned short*)d[1];
    const int N = (int)d[2], Cin = (int)d[3], taps = (int)d[4];
    int t = (int)(b - d[5]);
    const int tc = (Cin + 63) / 64, tn = (N + 63) / 64;
    const int c0 = (t % tc) * 64; t /= tc;
    const int n0 = (t % tn) * 64, tap = t / tn;
    src += (long)tap * Cin;
    dst += (long)(taps - 1 - tap) * N;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool pair_in = (Cin & 1) == 0 && (((uintptr_t)src) & 3) == 0, pair_out = (N & 1) == 0 && (((uintptr_t)dst) & 3) == 0;
    for (int r = ty; r < 64; r += 8) {
        const int nn = n0 + r, c = c0 + 2 * tx;
        if (nn >= N) continue;
        const unsigned short* s = src + (long)nn * taps * Cin + c;
        if (pair_in && c + 1 < Cin) {
            const unsigned v = *(const unsigned*)s;
            tile[r][2 * tx] = (unsigned short)v;
            tile[r][2 * tx + 1] = (unsigned short)(v >> 16);
        } else {
            if (c < Cin) tile[r][2 * tx] = s[0];
            if (c + 1 < Cin) tile[r][2 * tx + 1] = s[1];
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 8) {
        const int c = c0 + r, nn = n0 + 2 * tx;
        if (c >= Cin) continue;
        unsigned short* o = dst + (long)c * taps * N + nn;
        if (pair_out && nn + 1 < N) *(unsigned*)o = (unsigned)tile[2 * tx][r] | ((unsigned)tile[2 * tx + 1][r] << 16);
        else {
            if (nn < N) o[0] = tile[2 * tx][r];
            if (nn + 1 < N) o[1] = tile[2 * tx + 1][r];
        }
    }
}

// ---- depth-wise conv weight gradient -------------------------------------------------------------------------------
// Workgroup = DWG_R rows x 64 channels of one utterance: dy tile and the (DWG_R + k - 1)-row x slab in LDS (fp32), a thread
// owns one channel and every 4th row and keeps all k taps (+ the bias sum) in registers; the 4 row groups meet in LDS.
constexpr int DWG_R = 128, DWG_KMAX = 32;
template <typename T, int KB>  // KB = taps rounded up to 4 / 8 / 16 / 32: the tap loops are unrolled to KB
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwConvWgradArgs p) {
    __shared__ __attribute__((aligned(16))) float xs[(DWG_R + KB - 1) * 64];
    __shared__ __attribute__((aligned(16))) float ds[DWG_R * 64];
    const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
    const int t0 = blockIdx.x * DWG_R, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int nchunk = gridDim.x;
    const T* x = (const T*)p.x + (size_t)b * p.S * p.C;
    const T* dy = (const T*)p.dy + (size_t)b * p.S * p.C;
    const int rows = DWG_R + p.k - 1;
    const bool quad = (p.C & 3) == 0 && c0 + 64 <= p.C;  // 8- / 16-byte loads of four channels
    if (quad) {
        // every load of both tiles is issued before the first is used (clamped row, zeroed afterwards): one memory round trip
        // per workgroup instead of one per 16 rows
        constexpr int NX = ((DWG_R + KB - 1) * 16 + 255) / 256, ND = DWG_R * 16 / 256;
        float vx[NX][4], vd[ND][4];
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + u * 256, r = i >> 4, cq = (i & 15) * 4, t = t0 + r - p.pad;
            const int tc = t < 0 ? 0 : (t < p.S ? t : p.S - 1);
            ld4<T>(x + (size_t)tc * p.C + c0 + cq, vx[u]);
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + u * 256, r = i >> 4, cq = (i & 15) * 4, t = t0 + r;
            ld4<T>(dy + (size_t)(t < p.S ? t : p.S - 1) * p.C + c0 + cq, vd[u]);
        }
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + u * 256, r = i >> 4, cq = (i & 15) * 4, t = t0 + r - p.pad;
            const bool v = t >= 0 && t < p.S;
            if (r < rows) *(float4*)(xs + r * 64 + cq) = v ? make_float4(vx[u][0], vx[u][1], vx[u][2], vx[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + u * 256, r = i >> 4, cq = (i & 15) * 4, t = t0 + r;
            *(float4*)(ds + r * 64 + cq) = t < p.S ? make_float4(vd[u][0], vd[u][1], vd[u][2], vd[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int i = tid; i < rows * 64; i += 256) {
            const int r = i >> 6, cc = i & 63, t = t0 + r - p.pad;
            xs[i] = (t >= 0 && t < p.S && c0 + cc < p.C) ? Num<T>::to_f32(x[(size_t)t * p.C + c0 + cc]) : 0.f;
        }
        for (int i = tid; i < DWG_R * 64; i += 256) {
            const int r = i >> 6, cc = i & 63, t = t0 + r;
            ds[i] = (t < p.S && c0 + cc < p.C) ? Num<T>::to_f32(dy[(size_t)t * p.C + c0 + cc]) : 0.f;
        }
    }
    __syncthreads();
    float acc[KB], bsum = 0.f;
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[j] = 0.f;
    for (int r = g; r < DWG_R; r += 4) {
        const float d = ds[r * 64 + c];
        bsum += d;
#pragma unroll
        for (int j = 0; j < KB; ++j)
            if (j < p.k) acc[j] = fmaf(d, xs[(r + j) * 64 + c], acc[j]);
    }
    __syncthreads();
    // reduce the 4 row groups through LDS (reusing xs): [g][j][c], j = KB holds the bias sum
#pragma unroll
    for (int j = 0; j < KB; ++j) xs[(g * (KB + 1) + j) * 64 + c] = acc[j];
    xs[(g * (KB + 1) + KB) * 64 + c] = bsum;
    __syncthreads();
    float* part = p.part + ((size_t)b * nchunk + blockIdx.x) * (size_t)p.C * (p.k + 1);
    for (int i = tid; i < (p.k + 1) * 64; i += 256) {
        const int j = i >> 6, cc = i & 63;
        if (c0 + cc >= p.C) continue;
        const int jj = j < p.k ? j : KB;
        const float v = (xs[(0 * (KB + 1) + jj) * 64 + cc] + xs[(1 * (KB + 1) + jj) * 64 + cc]) +
                        (xs[(2 * (KB + 1) + jj) * 64 + cc] + xs[(3 * (KB + 1) + jj) * 64 + cc]);
        if (j < p.k) part[(size_t)(c0 + cc) * p.k + j] = v;
        else part[(size_t)p.C * p.k + c0 + cc] = v;
    }
}

template <typename T>
__global__ void fold_conv2_kernel(FoldConv2Args p) {
    const int gs = p.F / p.H;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)p.H * p.F) {
        const int o = (int)(i / p.F), f = (int)(i % p.F), g = f / gs, j = f % gs;
        float a = 0.f;
        for (int ii = 0; ii < gs; ++ii) a = fmaf(p.W21[(long)o * p.F + g * gs + ii], p.G[(long)(g * gs + ii) * gs + j], a);
        ((T*)p.Wf)[i] = Num<T>::from_f32(a);
    }
}
// bf[o] = b21[o] + W21[o] . bg : one wave per output row
__global__ __launch_bounds__(256) void fold_conv2_bias_kernel(FoldConv2Args p) {
    const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= p.H) return;
    float a = 0.f;
    for (int f = lane; f < p.F; f += 64) a = fmaf(p.W21[(long)o * p.F + f], p.bg[f], a);
    a = wave_sum(a);
    if (lane == 0) p.bf[o] = p.b21[o] + a;
}
__global__ void unfold_conv2_w_kernel(UnfoldConv2Args p) {  // dW21[o][f] += dbf[o] bg[f] + sum_j dWf[o][g*gs + j] G[f][j]
    const int gs = p.F / p.H;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)p.H * p.F) {
        const int o = (int)(i / p.F), f = (int)(i % p.F), g = f / gs;
        float a = p.dbf[o] * p.bg[f];
        for (int j = 0; j < gs; ++j) a = fmaf(p.dWf[(long)o * p.F + g * gs + j], p.G[(long)f * gs + j], a);
        p.dW21[i] += a;
    }
    if (i < p.H) p.db21[i] += p.dbf[i];
}
// dG[f][j] += sum_o W21[o][f] dWf[o][g*gs + j], dbg[f] += sum_o W21[o][f] dbf[o]: a workgroup per input channel f, the 256
// threads share the output rows, fixed-order LDS tree
constexpr int UF_GS = 8;  // group sizes up to 8 (F / H; 4 in every BASELINE config)
__global__ __launch_bounds__(256) void unfold_conv2_g_kernel(UnfoldConv2Args p) {
    __shared__ float red[256][UF_GS + 1];
    const int gs = p.F / p.H, f = blockIdx.x, g = f / gs, tid = threadIdx.x;
    float a[UF_GS + 1];
#pragma unroll
    for (int j = 0; j <= UF_GS; ++j) a[j] = 0.f;
    for (int o = tid; o < p.H; o += 256) {
        const float w = p.W21[(long)o * p.F + f];
#pragma unroll
        for (int j = 0; j < UF_GS; ++j)
            if (j < gs) a[j] = fmaf(w, p.dWf[(long)o * p.F + g * gs + j], a[j]);
        a[UF_GS] = fmaf(w, p.dbf[o], a[UF_GS]);
    }
#pragma unroll
    for (int j = 0; j <= UF_GS; ++j) red[tid][j] = a[j];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int j = 0; j <= UF_GS; ++j) red[tid][j] += red[tid + st][j];
        __syncthreads();
    }
    if (tid < gs) p.dG[(long)f * gs + tid] += red[0][tid];
    if (tid == 0) p.dbg[f] += red[0][UF_GS];
}

inline int ok() { return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP; }

}  // namespace

int launch_dropout(const DropoutArgs& a, int dtype, hipStream_t stream) {
    if (!(a.p >= 0.f && a.p < 1.f)) return FS2_ERR_ARG;
    if (!a.n) return FS2_OK;
    size_t blocks = (a.n + 2047) / 2048;
    if (blocks > 4096) blocks = 4096;
    if (dtype == FS2_BF16) hipLaunchKernelGGL(dropout_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (dtype == FS2_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else return FS2_ERR_SHAPE;
    return ok();
}
int launch_attn_delta(const AttnDeltaArgs& a, int dtype, hipStream_t stream) {
    if (a.B <= 0 || a.S <= 0 || a.heads <= 0 || a.H % a.heads) return FS2_ERR_SHAPE;
    const long n = (long)a.B * a.S * a.heads;
    const dim3 g((unsigned)((n + 3) / 4));
    if (dtype == FS2_BF16) hipLaunchKernelGGL(attn_delta_kernel<bf16>, g, dim3(256), 0, stream, a);
    else if (dtype == FS2_F32) hipLaunchKernelGGL(attn_delta_kernel<float>, g, dim3(256), 0, stream, a);
    else return FS2_ERR_SHAPE;
    return ok();
}
int launch_row_dot(const RowDotArgs& a, int dtype, hipStream_t stream) {
    if (a.M <= 0 || a.H <= 0) return FS2_ERR_SHAPE;
    const dim3 g((unsigned)((a.M + 3) / 4));
    if (dtype == FS2_BF16) hipLaunchKernelGGL(row_dot_kernel<bf16>, g, dim3(256), 0, stream, a);
    else if (dtype == FS2_F32) hipLaunchKernelGGL(row_dot_kernel<float>, g, dim3(256), 0, stream, a);
    else return FS2_ERR_SHAPE;
    return ok();
}

int dwconv_wgrad_parts(int B, int S) { return B * ((S + DWG_R - 1) / DWG_R); }
int launch_dwconv_wgrad(const DwConvWgradArgs& a, int dtype, hipStream_t stream) {
    if (a.k > DWG_KMAX || a.k < 1 || a.B <= 0 || a.S <= 0 || (dtype != FS2_BF16 && dtype != FS2_F32)) return FS2_ERR_SHAPE;
    const dim3 g((a.S + DWG_R - 1) / DWG_R, (a.C + 63) / 64, a.B);
#define FS2_DWG(KB) \
    do { \
        if (dtype == FS2_BF16) hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16, KB>), g, dim3(256), 0, stream, a); \
        else hipLaunchKernelGGL((dwconv_wgrad_kernel<float, KB>), g, dim3(256), 0, stream, a); \
    } while (0)
    if (a.k <= 4) FS2_DWG(4); else if (a.k <= 8) FS2_DWG(8); else if (a.k <= 16) FS2_DWG(16); else FS2_DWG(32);
#undef FS2_DWG
    return ok();
}
int launch_fold_conv2(const FoldConv2Args& a, int wf_dtype, hipStream_t stream) {
    if (a.H <= 0 || a.F % a.H) return FS2_ERR_SHAPE;
    const long n = (long)a.H * a.F;
    const dim3 g((unsigned)((n + 255) / 256));
    if (wf_dtype == FS2_BF16) hipLaunchKernelGGL(fold_conv2_kernel<bf16>, g, dim3(256), 0, stream, a);
    else if (wf_dtype == FS2_F32) hipLaunchKernelGGL(fold_conv2_kernel<float>, g, dim3(256), 0, stream, a);
    else return FS2_ERR_SHAPE;
    hipLaunchKernelGGL(fold_conv2_bias_kernel, dim3((a.H + 3) / 4), dim3(256), 0, stream, a);
    return ok();
}
int launch_unfold_conv2(const UnfoldConv2Args& a, hipStream_t stream) {
    if (a.H <= 0 || a.F % a.H || a.F / a.H > UF_GS) return FS2_ERR_SHAPE;
    const long n = (long)a.H * a.F;
    hipLaunchKernelGGL(unfold_conv2_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(unfold_conv2_g_kernel, dim3(a.F), dim3(256), 0, stream, a);
    return ok();
}

int launch_transpose_weight(const TransposeWeightArgs& a, int dtype, hipStream_t stream) {
    if (a.N <= 0 || a.Cin <= 0 || a.taps <= 0) return FS2_ERR_SHAPE;
    const dim3 g((a.Cin + 31) / 32, (a.N + 31) / 32, a.taps);
    if (dtype == FS2_BF16) hipLaunchKernelGGL(transpose_weight_kernel<bf16>, g, dim3(256), 0, stream, a);
    else if (dtype == FS2_F32) hipLaunchKernelGGL(transpose_weight_kernel<float>, g, dim3(256), 0, stream, a);
    else return FS2_ERR_SHAPE;
    return ok();
}

int launch_transpose_weight_batch(const long long* tab, int n, long long tiles, hipStream_t stream) {
    if (!tab || n <= 0 || tiles <= 0 || tiles > 0x7fffffffLL) return FS2_ERR_ARG;
    hipLaunchKernelGGL(transpose_weight_batch_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, tab, n);
    return ok();
}

int layernorm_bwd_parts(int M) { return (M + LN_ROWS - 1) / LN_ROWS; }

int launch_layernorm_bwd(const LayerNormBwdArgs& a, int dtype, hipStream_t stream) {
    if ((dtype != FS2_F32 && dtype != FS2_BF16) || a.H > 256 * LN_MAXC || a.M <= 0) return FS2_ERR_SHAPE;
    if (a.nparts != layernorm_bwd_parts(a.M) || a.H % 4) return FS2_ERR_ARG;
    const int nc = (a.H + 255) / 256;
#define FS2_LNB(T, NC) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NC>), dim3(a.nparts), dim3(256), 0, stream, a)
    if (dtype == FS2_F32) { if (nc == 1) FS2_LNB(float, 1); else if (nc == 2) FS2_LNB(float, 2); else if (nc == 3) FS2_LNB(float, 3); else FS2_LNB(float, 4); }
    else { if (nc == 1) FS2_LNB(bf16, 1); else if (nc == 2) FS2_LNB(bf16, 2); else if (nc == 3) FS2_LNB(bf16, 3); else FS2_LNB(bf16, 4); }
#undef FS2_LNB
    return ok();
}

// A/B knob: one-launch column sums (last workgroup reduces) / two launches.  OFF: measured on MI355X the one-launch form costs
// ~50 us per launch more than it saves (C2 training step 13.5 -> 18.0 ms over its 90 column sums): each workgroup's device-scope
// __threadfence() is an L2 write-back + invalidate on this 8-XCD part, thousands of them per launch.
// (Tuning::colsum_fused, default 0)
static int cs_chunks(int M, int seg) { return ((seg > 0 ? seg : M) + CS_CHUNK - 1) / CS_CHUNK; }
size_t col_sum_ws_bytes(int M, int N, int seg) {
    const int nseg = seg > 0 ? M / seg : 1;
    return (size_t)CS_CTR * sizeof(unsigned) + (size_t)nseg * cs_chunks(M, seg) * N * sizeof(float);
}
int launch_col_sum(const ColSumArgs& a, int dtype, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || (a.seg > 0 && a.M % a.seg) || !a.ws) return FS2_ERR_SHAPE;
    if (a.out2 && (a.n1 <= 0 || a.n1 >= a.N)) return FS2_ERR_ARG;
    const int nseg = a.seg > 0 ? a.M / a.seg : 1, nchunk = cs_chunks(a.M, a.seg);
    const dim3 g1((a.N + 63) / 64, nchunk, nseg);
    const bool fused = tuning_of(a.tune).colsum_fused && (long)g1.x * nseg <= CS_CTR;
    if (fused) {
        if (dtype == FS2_BF16) hipLaunchKernelGGL((col_sum_pass1<bf16, true>), g1, dim3(256), 0, stream, a, nchunk);
        else hipLaunchKernelGGL((col_sum_pass1<float, true>), g1, dim3(256), 0, stream, a, nchunk);
        return ok();
    }
    if (dtype == FS2_BF16) hipLaunchKernelGGL((col_sum_pass1<bf16, false>), g1, dim3(256), 0, stream, a, nchunk);
    else hipLaunchKernelGGL((col_sum_pass1<float, false>), g1, dim3(256), 0, stream, a, nchunk);
    if (nchunk > 1) hipLaunchKernelGGL(col_sum_pass2, dim3((a.N + 63) / 64, nseg), dim3(256), 0, stream, a, nchunk);
    return ok();
}

int launch_softmax_fwd(const SoftmaxArgs& a, int dtype, hipStream_t stream) {
    if (dtype != FS2_F32 && dtype != FS2_BF16) return FS2_ERR_SHAPE;
    const long rows = (long)a.B * a.heads * a.S;
    const dim3 g((unsigned)((rows + 3) / 4));
    const bool row_ok = a.S % 4 == 0 && a.S <= 2048 && ((uintptr_t)a.s & 15) == 0 && ((uintptr_t)a.out & 7) == 0;
#define FS2_SMF(T, NV) hipLaunchKernelGGL((softmax_fwd_row_kernel<T, NV>), g, dim3(256), 0, stream, a)
    if (row_ok && dtype == FS2_F32) { if (a.S <= 256) FS2_SMF(float, 1); else if (a.S <= 512) FS2_SMF(float, 2); else if (a.S <= 1024) FS2_SMF(float, 4); else FS2_SMF(float, 8); }
    else if (row_ok) { if (a.S <= 256) FS2_SMF(bf16, 1); else if (a.S <= 512) FS2_SMF(bf16, 2); else if (a.S <= 1024) FS2_SMF(bf16, 4); else FS2_SMF(bf16, 8); }
#undef FS2_SMF
    else if (dtype == FS2_F32) hipLaunchKernelGGL(softmax_fwd_kernel<float>, g, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(softmax_fwd_kernel<bf16>, g, dim3(256), 0, stream, a);
    return ok();
}
int launch_softmax_bwd(const SoftmaxArgs& a, int dtype, hipStream_t stream) {
    if (dtype != FS2_F32 && dtype != FS2_BF16) return FS2_ERR_SHAPE;
    const long rows = (long)a.B * a.heads * a.S;
    const dim3 g((unsigned)((rows + 3) / 4));
    const bool row_ok = a.S % 4 == 0 && a.S <= 2048 && ((uintptr_t)a.s & 15) == 0 && ((uintptr_t)a.out & 7) == 0 && ((uintptr_t)a.p & 7) == 0;
#define FS2_SMB(T, NV) hipLaunchKernelGGL((softmax_bwd_row_kernel<T, NV>), g, dim3(256), 0, stream, a)
    if (row_ok && dtype == FS2_F32) { if (a.S <= 256) FS2_SMB(float, 1); else if (a.S <= 512) FS2_SMB(float, 2); else if (a.S <= 1024) FS2_SMB(float, 4); else FS2_SMB(float, 8); }
    else if (row_ok) { if (a.S <= 256) FS2_SMB(bf16, 1); else if (a.S <= 512) FS2_SMB(bf16, 2); else if (a.S <= 1024) FS2_SMB(bf16, 4); else FS2_SMB(bf16, 8); }
#undef FS2_SMB
    else if (dtype == FS2_F32) hipLaunchKernelGGL(softmax_bwd_kernel<float>, g, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(softmax_bwd_kernel<bf16>, g, dim3(256), 0, stream, a);
    return ok();
}

int launch_ew(const EwArgs& a0, int dtype, hipStream_t stream) {
    if (!a0.n) return FS2_OK;
    EwArgs a = a0;
    const size_t ne = dtype == FS2_BF16 ? 8 : 4, esz = dtype == FS2_BF16 ? 2 : 4;
    const bool al = (((uintptr_t)a.a | (uintptr_t)a.b | (uintptr_t)a.out) & 15) == 0;
    if (al && a.n >= ne) {
        EwArgs v = a;
        v.n = a.n / ne * ne;
        size_t blocks = (v.n / ne + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (dtype == FS2_BF16) hipLaunchKernelGGL(ew_vec_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, v);
        else hipLaunchKernelGGL(ew_vec_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, v);
        a.a = (const char*)a.a + v.n * esz;
        if (a.b) a.b = (const char*)a.b + v.n * esz;
        a.out = (char*)a.out + v.n * esz;
        a.n -= v.n;
        if (!a.n) return ok();
    }
    size_t blocks = (a.n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    if (dtype == FS2_BF16) hipLaunchKernelGGL(ew_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(ew_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return ok();
}

static bool scatter_chunked(int R, int V) { return V <= SR_VMAX && R >= 4 * SR_CHUNK; }
size_t scatter_rows_ws_bytes(int R, int H, int V) {
    if (!scatter_chunked(R, V)) return 0;
    const int nchunk = (R + SR_CHUNK - 1) / SR_CHUNK;
    return (size_t)nchunk * V * H * sizeof(float) + col_sum_ws_bytes(nchunk, V * H, 0);
}
int launch_scatter_rows(const ScatterRowsArgs& a, int dtype, hipStream_t stream) {
    if (a.R <= 0 || a.V <= 0 || (!a.idx32 && !a.idx64)) return FS2_ERR_ARG;
    if (a.ws && scatter_chunked(a.R, a.V)) {
        const int nchunk = (a.R + SR_CHUNK - 1) / SR_CHUNK;
        const dim3 g(nchunk, (a.H + 63) / 64);
        if (dtype == FS2_BF16) hipLaunchKernelGGL(scatter_rows_part_kernel<bf16>, g, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(scatter_rows_part_kernel<float>, g, dim3(256), 0, stream, a);
        ColSumArgs cs{a.ws, a.table, a.ws + (size_t)nchunk * a.V * a.H, nchunk, a.V * a.H, a.V * a.H, 0, 1, 1.f};
        return launch_col_sum(cs, FS2_F32, stream);
    }
    if (dtype == FS2_BF16) hipLaunchKernelGGL(scatter_rows_kernel<bf16>, dim3(a.V), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(scatter_rows_kernel<float>, dim3(a.V), dim3(256), 0, stream, a);
    return ok();
}

int launch_regulate_bwd(const RegulateBwdArgs& a, int dtype, hipStream_t stream) {
    const long n = (long)a.B * a.L;
    const dim3 g((unsigned)((n + 3) / 4));
    if (dtype == FS2_BF16) hipLaunchKernelGGL(regulate_bwd_kernel<bf16>, g, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(regulate_bwd_kernel<float>, g, dim3(256), 0, stream, a);
    return ok();
}

int launch_masked_loss_bwd(const LossBwdArgs& a, hipStream_t stream) {
    const int64_t total = a.rows * (int64_t)a.inner;
    int64_t blocks = (total + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(masked_loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return ok();
}

size_t sum_sq_ws_bytes(size_t) { return SS_BLOCKS * sizeof(double); }
int launch_sum_sq(const float* x, size_t n, float* ws, float* out, hipStream_t stream) {
    int nb = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (nb < 1) nb = 1;
    if (nb > SS_BLOCKS) nb = SS_BLOCKS;
    hipLaunchKernelGGL(sum_sq_pass1, dim3(nb), dim3(256), 0, stream, x, n, ws);
    hipLaunchKernelGGL(sum_sq_pass2, dim3(1), dim3(64), 0, stream, ws, nb, out);
    return ok();
}

int launch_adamw(const AdamWArgs& a, hipStream_t stream) {
    if (!a.n || a.step < 1) return FS2_ERR_ARG;
    size_t blocks = (a.n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return ok();
}

}  // namespace fs2
