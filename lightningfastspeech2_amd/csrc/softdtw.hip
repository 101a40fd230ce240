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

size_t soft_dtw_lds_bytes(int N, int M, int D) { return (size_t)3 * (N + 2) * 8 + (size_t)(N + M) * (D + 1) * 4; }

int launch_soft_dtw(const SoftDtwArgs& a, hipStream_t stream) {
    if (a.B <= 0) return FS2_OK;
    if (a.N <= 0 || a.M <= 0 || a.D <= 0 || !(a.gamma > 0.f)) return FS2_ERR_ARG;
    size_t lds = soft_dtw_lds_bytes(a.N, a.M, a.D);
    const bool stage = lds <= 160 * 1024;
    if (!stage) lds = (size_t)3 * (a.N + 2) * 8;
    if (lds > 160 * 1024) return FS2_ERR_SHAPE;  // the three diagonals must fit one CU's LDS (N <= 6800)
    static bool attr = false;
    if (lds > 64 * 1024 && !attr) {
        if (hipFuncSetAttribute((const void*)soft_dtw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)soft_dtw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return FS2_ERR_HIP;
        attr = true;
    }
    if (stage) hipLaunchKernelGGL(soft_dtw_kernel<true>, dim3(a.B), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(soft_dtw_kernel<false>, dim3(a.B), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
