// Soft-DTW value (forward) on gfx950: the validation metric of the reference (fastspeech2.py:1149-1156 ->
// litfass/third_party/softdtw/__init__.py:8-24,110-139) and the arithmetic of its "soft_dtw" loss kind (loss.py:57-81).
//   D[i,j] = sum_d (x[i,d] - y[j,d])^2  (fp32, as torch computes it);  R[0,0] = 0, infinite border;
//   R[i,j] = D[i-1,j-1] - gamma * log(exp(-R[i-1,j-1]/gamma) + exp(-R[i-1,j]/gamma) + exp(-R[i,j-1]/gamma))   (fp64)
//   value = R[N,M] (returned as fp32, like the reference).
// One workgroup per sequence pair.  The recursion only couples a cell to the two previous anti-diagonals, so three
// diagonals of fp64 (indexed by i, N + 2 entries each) rotate through LDS and the cells of a diagonal are computed by the
// 256 threads in parallel, one barrier per diagonal.  Both sequences are staged in LDS as fp32 once (rows of D + 1 floats:
// lane-stride accesses fall on different banks), so a cell's distance is 2 * D LDS reads, no global traffic in the loop.
// A latency-bound dynamic program (N + M - 1 dependent steps), not a bandwidth kernel: what matters is that a step is short.
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

template <bool STAGE>  // STAGE: both sequences fit in LDS next to the diagonals; else they are read through L1 / L2
__global__ __launch_bounds__(256) void soft_dtw_kernel(SoftDtwArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = p.N, M = p.M, D = p.D, ld = STAGE ? D + 1 : D;
    double* r0 = (double*)smem;            // diagonal d - 2
    double* r1 = r0 + (N + 2);             // diagonal d - 1
    double* r2 = r1 + (N + 2);             // diagonal d
    const float* x = p.x + (size_t)b * N * D;
    const float* y = p.y + (size_t)b * M * D;
    const float *xs = x, *ys = y;
    if constexpr (STAGE) {
        float* xl = (float*)(r2 + (N + 2));    // [N][D + 1]
        float* yl = xl + (size_t)N * ld;       // [M][D + 1]
        for (int i = tid; i < N * D; i += 256) xl[(i / D) * ld + i % D] = x[i];
        for (int i = tid; i < M * D; i += 256) yl[(i / D) * ld + i % D] = y[i];
        xs = xl;
        ys = yl;
    }
    const double inf = __builtin_inf();
    for (int i = tid; i < N + 2; i += 256) {
        r0[i] = i == 0 ? 0.0 : inf;   // diagonal 0: R[0,0] = 0
        r1[i] = inf;                  // diagonal 1: R[0,1] = R[1,0] = inf
        r2[i] = inf;
    }
    __syncthreads();
    const double gamma = (double)p.gamma, ig = -1.0 / gamma;
    for (int d = 2; d <= N + M; ++d) {
        const int lo = d - M > 1 ? d - M : 1, hi = d - 1 < N ? d - 1 : N;
        for (int i = lo + tid; i <= hi; i += 256) {
            const int j = d - i;
            const float* xi = xs + (size_t)(i - 1) * ld;
            const float* yj = ys + (size_t)(j - 1) * ld;
            float dist = 0.f;
            for (int c = 0; c < D; ++c) { const float t = xi[c] - yj[c]; dist = fmaf(t, t, dist); }
            const double a0 = r0[i - 1] * ig, a1 = r1[i - 1] * ig, a2 = r1[i] * ig;  // -R[i-1,j-1]/g, -R[i-1,j]/g, -R[i,j-1]/g
            const double mx = fmax(fmax(a0, a1), a2);
            const double sm = exp(a0 - mx) + exp(a1 - mx) + exp(a2 - mx);
            r2[i] = (double)dist - gamma * (log(sm) + mx);
        }
        if (tid == 0) {  // the border of this diagonal: R[0, d] and (if it exists) R[d, 0]
            r2[0] = inf;
            if (d <= N + 1) r2[d] = inf;
        }
        __syncthreads();
        double* t = r0; r0 = r1; r1 = r2; r2 = t;
    }
    if (tid == 0) p.out[b] = (float)r1[N];  // after the last rotation r1 holds diagonal N + M
}

// Soft-DTW value AND its gradient with respect to x: what loss.backward() computes through the reference's vendored module
// (third_party/softdtw/__init__.py:27-52 compute_softdtw_backward, :56-77 _SoftDTW, :85-92 calc_distance_matrix):
//   forward as above, with R kept - as float32, which is what _SoftDTW.forward saves ("torch.Tensor(R).type(dtype)", :63) - and
//   the distances D (float32);  backward  E[N,M] = 1,
//   E[i,j] = E[i+1,j] a + E[i,j+1] b + E[i+1,j+1] c,  a = exp((R[i+1,j] - R[i,j] - D[i+1,j]) / gamma) etc. in fp64 from those
//   float32 values (R's far border -inf, R[N+1,M+1] = R[N,M], D's border 0), E handed back as float32 (:75);
//   d value / d x[i,:] = 2 sum_j E[i,j] (x[i,:] - y[j,:])   (autograd through pow(x - y, 2).sum(3)).
// One workgroup per pair: the forward's anti-diagonals, then the same walk backwards with three rotating E diagonals in LDS
// (R, D of the three neighbour cells come back from the scratch: L2), then the contraction with one thread per (i, channel).
template <bool STAGE>
__global__ __launch_bounds__(256) void soft_dtw_grad_kernel(SoftDtwArgs p, float* __restrict__ gx, float* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = p.N, M = p.M, D = p.D, ld = STAGE ? D + 1 : D;
    double* r0 = (double*)smem;
    double* r1 = r0 + (N + 2);
    double* r2 = r1 + (N + 2);
    const float* x = p.x + (size_t)b * N * D;
    const float* y = p.y + (size_t)b * M * D;
    const float *xs = x, *ys = y;
    if constexpr (STAGE) {
        float* xl = (float*)(r2 + (N + 2));
        float* yl = xl + (size_t)N * ld;
        for (int i = tid; i < N * D; i += 256) xl[(i / D) * ld + i % D] = x[i];
        for (int i = tid; i < M * D; i += 256) yl[(i / D) * ld + i % D] = y[i];
        xs = xl;
        ys = yl;
    }
    // scratch of this pair: R (N+2) x (M+2) float32, D N x M float32, E N x M float32
    const size_t RS = (size_t)(N + 2) * (M + 2), NM = (size_t)N * M;
    float* Rg = scratch + (size_t)b * (RS + 2 * NM);
    float* Dg = Rg + RS;
    float* Eg = Dg + NM;
    const int W = M + 2;
    const double inf = __builtin_inf();
    for (int i = tid; i < N + 2; i += 256) {
        r0[i] = i == 0 ? 0.0 : inf;
        r1[i] = inf;
        r2[i] = inf;
    }
    // the far border of R as the backward pass wants it (:37-39): column M+1 and row N+1 = -inf, the corner = R[N,M] (set below)
    for (int i = tid; i < N + 2; i += 256) Rg[(size_t)i * W + (M + 1)] = -__builtin_inff();
    for (int j = tid; j < M + 2; j += 256) Rg[(size_t)(N + 1) * W + j] = -__builtin_inff();
    __syncthreads();
    const double gamma = (double)p.gamma, ig = -1.0 / gamma;
    for (int d = 2; d <= N + M; ++d) {
        const int lo = d - M > 1 ? d - M : 1, hi = d - 1 < N ? d - 1 : N;
        for (int i = lo + tid; i <= hi; i += 256) {
            const int j = d - i;
            const float* xi = xs + (size_t)(i - 1) * ld;
            const float* yj = ys + (size_t)(j - 1) * ld;
            float dist = 0.f;
            for (int c = 0; c < D; ++c) { const float t = xi[c] - yj[c]; dist = fmaf(t, t, dist); }
            const double a0 = r0[i - 1] * ig, a1 = r1[i - 1] * ig, a2 = r1[i] * ig;
            const double mx = fmax(fmax(a0, a1), a2);
            const double sm = exp(a0 - mx) + exp(a1 - mx) + exp(a2 - mx);
            const double r = (double)dist - gamma * (log(sm) + mx);
            r2[i] = r;
            Rg[(size_t)i * W + j] = (float)r;
            Dg[(size_t)(i - 1) * M + (j - 1)] = dist;
        }
        if (tid == 0) {
            r2[0] = inf;
            if (d <= N + 1) r2[d] = inf;
        }
        __syncthreads();
        double* t = r0; r0 = r1; r1 = r2; r2 = t;
    }
    if (tid == 0) {
        p.out[b] = (float)r1[N];
        Rg[(size_t)(N + 1) * W + (M + 1)] = (float)r1[N];  // R[:, -1, -1] = R[:, -2, -2]
    }
    __threadfence_block();
    __syncthreads();  // this workgroup's R / D stores are visible to its own later loads

    // ---- backward: E diagonals d = N + M + 2 (the corner, E = 1) down to 2; e0 = diagonal d + 2, e1 = d + 1, e2 = d, indexed by i
    double *e0 = r0, *e1 = r1, *e2 = r2;
    for (int i = tid; i < N + 2; i += 256) {
        e0[i] = i == N + 1 ? 1.0 : 0.0;   // diagonal N + M + 2 holds only (N+1, M+1)
        e1[i] = 0.0;                      // diagonal N + M + 1: (N+1, M), (N, M+1): border zeros
        e2[i] = 0.0;
    }
    __syncthreads();
    const double rg = 1.0 / gamma;
    for (int d = N + M; d >= 2; --d) {
        const int lo = d - M > 1 ? d - M : 1, hi = d - 1 < N ? d - 1 : N;
        for (int i = lo + tid; i <= hi; i += 256) {
            const int j = d - i;
            const double rij = (double)Rg[(size_t)i * W + j];
            auto dd = [&](int ii, int jj) -> double { return (ii <= N && jj <= M) ? (double)Dg[(size_t)(ii - 1) * M + (jj - 1)] : 0.0; };
            const double a = exp(((double)Rg[(size_t)(i + 1) * W + j] - rij - dd(i + 1, j)) * rg);
            const double bb = exp(((double)Rg[(size_t)i * W + (j + 1)] - rij - dd(i, j + 1)) * rg);
            const double c = exp(((double)Rg[(size_t)(i + 1) * W + (j + 1)] - rij - dd(i + 1, j + 1)) * rg);
            // (i+1, j) and (i, j+1) lie on diagonal d + 1 (index i+1 / i), (i+1, j+1) on diagonal d + 2 (index i+1)
            const double e = e1[i + 1] * a + e1[i] * bb + e0[i + 1] * c;
            e2[i] = e;
            Eg[(size_t)(i - 1) * M + (j - 1)] = (float)e;
        }
        if (tid == 0) {  // border cells of this diagonal: (0, d) never read; (d - M - 1 .. ) handled by the zero fill: refresh the two ends
            if (hi + 1 <= N + 1) e2[hi + 1] = 0.0;
            if (lo - 1 >= 0) e2[lo - 1] = 0.0;
        }
        __syncthreads();
        double* t = e0; e0 = e1; e1 = e2; e2 = t;
    }
    __threadfence_block();
    __syncthreads();
    // ---- d value / d x[i, c] = 2 sum_j E[i, j] (x[i, c] - y[j, c]), fp32 like the reference's autograd
    float* gxb = gx + (size_t)b * N * D;
    for (int o = tid; o < N * D; o += 256) {
        const int i = o / D, c = o - i * D;
        const float xi = xs[(size_t)i * ld + c];
        const float* er = Eg + (size_t)i * M;
        float acc = 0.f;
        for (int j = 0; j < M; ++j) acc = fmaf(er[j], xi - ys[(size_t)j * ld + c], acc);
        gxb[o] = 2.f * acc;
    }
}

size_t soft_dtw_grad_scratch_bytes(int B, int N, int M) {
    return (size_t)B * ((size_t)(N + 2) * (M + 2) + 2 * (size_t)N * M) * sizeof(float);
}

static int soft_dtw_lds_setup(size_t lds);

int launch_soft_dtw_grad(const SoftDtwArgs& a, float* gx, void* scratch, size_t scratch_bytes, hipStream_t stream) {
    if (a.B <= 0) return FS2_OK;
    if (a.N <= 0 || a.M <= 0 || a.D <= 0 || !(a.gamma > 0.f) || !gx || !scratch) return FS2_ERR_ARG;
    if (scratch_bytes < soft_dtw_grad_scratch_bytes(a.B, a.N, a.M)) return FS2_ERR_ARG;
    size_t lds = soft_dtw_lds_bytes(a.N, a.M, a.D);
    const bool stage = lds <= 160 * 1024;
    if (!stage) lds = (size_t)3 * (a.N + 2) * 8;
    if (lds > 160 * 1024) return FS2_ERR_SHAPE;
    if (const int st = soft_dtw_lds_setup(lds)) return st;
    if (stage) hipLaunchKernelGGL(soft_dtw_grad_kernel<true>, dim3(a.B), dim3(256), lds, stream, a, gx, (float*)scratch);
    else hipLaunchKernelGGL(soft_dtw_grad_kernel<false>, dim3(a.B), dim3(256), lds, stream, a, gx, (float*)scratch);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

size_t soft_dtw_lds_bytes(int N, int M, int D) { return (size_t)3 * (N + 2) * 8 + (size_t)(N + M) * (D + 1) * 4; }

static int soft_dtw_lds_setup(size_t lds) {  // more than 64 KiB of dynamic LDS needs the attribute, once per kernel
    static bool attr = false;
    if (lds > 64 * 1024 && !attr) {
        const void* ks[4] = {(const void*)soft_dtw_kernel<true>, (const void*)soft_dtw_kernel<false>,
                             (const void*)soft_dtw_grad_kernel<true>, (const void*)soft_dtw_grad_kernel<false>};
        for (const void* k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FS2_ERR_HIP;
        attr = true;
    }
    return FS2_OK;
}

int launch_soft_dtw(const SoftDtwArgs& a, hipStream_t stream) {
    if (a.B <= 0) return FS2_OK;
    if (a.N <= 0 || a.M <= 0 || a.D <= 0 || !(a.gamma > 0.f)) return FS2_ERR_ARG;
    size_t lds = soft_dtw_lds_bytes(a.N, a.M, a.D);
    const bool stage = lds <= 160 * 1024;
    if (!stage) lds = (size_t)3 * (a.N + 2) * 8;
    if (lds > 160 * 1024) return FS2_ERR_SHAPE;  // the three diagonals must fit one CU's LDS (N <= 6800)
    if (const int st = soft_dtw_lds_setup(lds)) return st;
    if (stage) hipLaunchKernelGGL(soft_dtw_kernel<true>, dim3(a.B), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(soft_dtw_kernel<false>, dim3(a.B), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
