// Masked mean losses of FastSpeech2Loss (reference: litfass/fastspeech2/loss.py:57-81 get_loss,
// :83-213 forward): mean over the NON-padded rows of |pred - truth| ("l1") or (pred - truth)^2 ("mse"),
// rows = (utterance, position), `inner` values per row (80 mel bins, 1 for durations / variances).  The
// reference does masked_select + nn.L1Loss / nn.MSELoss; here one pass reads pred, truth and the pad mask
// once (HBM-bound: 2 * 4 bytes per element + 1 byte per row), accumulates in fp64 per thread, reduces a
// block through LDS in a fixed order, and the LAST block to finish adds the block partials in index
// order - so the result does not depend on scheduling (no float atomics).
#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_MAX_BLOCKS = 1024;

// ws layout: double partial_sum[LOSS_MAX_BLOCKS], double partial_cnt[LOSS_MAX_BLOCKS], unsigned done
__global__ __launch_bounds__(LOSS_THREADS) void masked_loss_kernel(LossArgs p) {
    __shared__ double ssum[LOSS_THREADS], scnt[LOSS_THREADS];
    __shared__ bool is_last;
    const int tid = threadIdx.x;
    double sum = 0.0, cnt = 0.0;
    // a thread walks (row, column) pairs with a fixed column stride: one 64-bit division per thread, none per element
    const int64_t total = p.rows * (int64_t)p.inner;
    const int64_t stride = (int64_t)gridDim.x * LOSS_THREADS;
    int64_t e = (int64_t)blockIdx.x * LOSS_THREADS + tid;
    int64_t r = e / p.inner;
    int c = (int)(e - r * p.inner);
    const int64_t dr = stride / p.inner;
    const int dc = (int)(stride - dr * p.inner);
    float fsum = 0.f;
    int fcnt = 0, run = 0;
    for (; e < total; e += stride) {
        if (!p.mask[r]) {  // True = pad
            float t;
            if (p.truth_kind == 0) t = ((const float*)p.truth)[e];
            else t = logf((float)((const int64_t*)p.truth)[e] + 1.0f);  // duration target: log(d + 1), loss.py:176
            const float d = p.pred[e] - t;
            fsum += p.kind == 0 ? fabsf(d) : d * d;
            ++fcnt;
        }
        if (++run == 64) {  // fp32 runs of at most 64 terms, fp64 across runs
            sum += (double)fsum; cnt += (double)fcnt;
            fsum = 0.f; fcnt = 0; run = 0;
        }
        r += dr;
        c += dc;
        if (c >= p.inner) { c -= p.inner; ++r; }
    }
    sum += (double)fsum;
    cnt += (double)fcnt;
    ssum[tid] = sum;
    scnt[tid] = cnt;
    __syncthreads();
    for (int s = LOSS_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            ssum[tid] += ssum[tid + s];
            scnt[tid] += scnt[tid + s];
        }
        __syncthreads();
    }
    double* psum = (double*)p.ws;
    double* pcnt = psum + LOSS_MAX_BLOCKS;
    unsigned* done = (unsigned*)(pcnt + LOSS_MAX_BLOCKS);
    if (tid == 0) {
        psum[blockIdx.x] = ssum[0];
        pcnt[blockIdx.x] = scnt[0];
        __threadfence();
        is_last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // the last block adds the block partials: each thread a fixed subset in index order, then the same LDS tree
    double ts = 0.0, tc = 0.0;
    for (unsigned b = tid; b < gridDim.x; b += LOSS_THREADS) {
        ts += ((volatile double*)psum)[b];
        tc += ((volatile double*)pcnt)[b];
    }
    ssum[tid] = ts;
    scnt[tid] = tc;
    __syncthreads();
    for (int st = LOSS_THREADS / 2; st > 0; st >>= 1) {
        if (tid < st) {
            ssum[tid] += ssum[tid + st];
            scnt[tid] += scnt[tid + st];
        }
        __syncthreads();
    }
    if (tid == 0) {
        p.out[0] = scnt[0] > 0.0 ? (float)(ssum[0] / scnt[0]) : __builtin_nanf("");  // torch: mean of an empty selection is nan
        p.out[1] = (float)scnt[0];
        *done = 0u;  // ready for the next launch on this workspace
    }
}

size_t masked_loss_ws_bytes() { return sizeof(double) * 2 * LOSS_MAX_BLOCKS + 64; }

int launch_masked_loss(const LossArgs& a, hipStream_t stream) {
    if (a.rows <= 0 || a.inner <= 0) return FS2_ERR_SHAPE;
    const int64_t total = a.rows * (int64_t)a.inner;
    int64_t blocks = (total + LOSS_THREADS * 8 - 1) / (LOSS_THREADS * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
    hipLaunchKernelGGL(masked_loss_kernel, dim3((unsigned)blocks), dim3(LOSS_THREADS), 0, stream, a);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2
