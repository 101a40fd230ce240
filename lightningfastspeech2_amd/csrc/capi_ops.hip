// Single-operator entry points of the C ABI (include/fs2.h): each wraps one launcher so the
// parity tests can compare every kernel against the CPU oracle's corresponding op in isolation.
#include <math.h>

#include "fs2_common.h"
#include "fs2_kernels.h"

using namespace fs2;

namespace fs2 {

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ s, D* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        d[i] = Num<D>::from_f32(Num<S>::to_f32(s[i]));
}

int launch_convert(const ConvertArgs& a, int sdt, int ddt, hipStream_t st) {
    if (a.n == 0) return FS2_OK;
    size_t blocks = (a.n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 g((unsigned)blocks), b(256);
    if (sdt == FS2_F32 && ddt == FS2_F32) hipLaunchKernelGGL((convert_kernel<float, float>), g, b, 0, st, (const float*)a.src, (float*)a.dst, a.n);
    else if (sdt == FS2_F32 && ddt == FS2_BF16) hipLaunchKernelGGL((convert_kernel<float, bf16>), g, b, 0, st, (const float*)a.src, (bf16*)a.dst, a.n);
    else if (sdt == FS2_BF16 && ddt == FS2_F32) hipLaunchKernelGGL((convert_kernel<bf16, float>), g, b, 0, st, (const bf16*)a.src, (float*)a.dst, a.n);
    else if (sdt == FS2_BF16 && ddt == FS2_BF16) hipLaunchKernelGGL((convert_kernel<bf16, bf16>), g, b, 0, st, (const bf16*)a.src, (bf16*)a.dst, a.n);
    else return FS2_ERR_ARG;
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2

// A/B switches: one value of fs2_op_set_gemm_variant / fs2_set_tuning applied to a Tuning (fs2_kernels.h).  Every value is listed: an
// undefined one is FS2_ERR_ARG (a typo in FS2_GEMM_KNOBS must not pass silently); every accepted call bumps Tuning::gen, which is part
// of an engine's hipGraph keys - a captured phase is never replayed with kernels chosen under other switches.
namespace fs2 {
int apply_knob(Tuning& t, int variant) {
    int ok = 1;
    if (variant == 1500 || variant == 1501) t.attn_x3 = variant - 1500;               // fp32-storage split modes: attention on fp32 MFMA / on bf16 x 3 split products (default)
    else if (variant >= 1400 && variant <= 1402) t.gemm_wres = variant - 1400;        // bf16 K = 256 plain GEMMs: slab kernel / weight-resident kernel where it pays (default) / wherever it applies
    else if (variant == 1340 || variant == 1341) t.enc_attn_out = variant - 1340;     // engine, bf16, H = 256, 2 heads: encoder attention + out-projection + LayerNorm as two launches / one (default)
    else if (variant == 1320 || variant == 1321) t.pred_fuse_embed = variant - 1320;  // engine: variance encoder (bucketize + embedding add) as the tail of its predictor launch: off / on (default)
    else if (variant >= 1200 && variant <= 1204) t.attn_pipe = variant - 1200;        // 1200: attention.hip only; 1201 / 1202 / 1204: the software-pipelined kernel with 32 / 64 / 96 queries per wave where it applies; 1203: by size (default)
    else if (variant == 1100 || variant == 1101) t.colsum_fused = variant - 1100;     // column sums in two launches (default) / one
    else if (variant == 1000 || variant == 1001) t.bgemm_tn256 = variant - 1000;      // 256 x 256 LDS-DMA kernel for eligible bf16 TN products off / on (default)
    else if (variant >= 905 && variant <= 908) { if (variant >= 907) t.attn_bwd_nb_dq = 3 + ((variant - 905) & 1); else t.attn_bwd_nb = 3 + ((variant - 905) & 1); }  // 905 / 906: dK,dV launch 3 / 4 blocks per wave; 907 / 908: the dQ launch
    else if (variant == 909) t.attn_bwd_nb = 1;                                         // dK,dV launch back to its default (1 block per wave)
    else if (variant >= 900 && variant <= 903) { if (variant >= 902) t.attn_bwd_nb_dq = ((variant - 900) & 1) + 1; else t.attn_bwd_nb = ((variant - 900) & 1) + 1; }  // 900 / 901: dK,dV launch 1 / 2 blocks per wave; 902 / 903: the dQ launch
    else if (variant == 904) t.attn_bwd_nb_dq = 0;                                      // the dQ launch by size (default)
    else if (variant == 800 || variant == 801) t.bgemm_full = variant - 800;           // bf16 strided-batched GEMM: generic instantiation only / bounds-free one for full aligned tiles (default)
    else if (variant == 700 || variant == 701) t.bgemm_xcd = variant - 700;            // bf16 strided-batched GEMM tile order plain / XCD-contiguous (default)
    else if (variant == 500 || variant == 501) t.split_f32 = variant - 500;            // operator level: fp32 slab launches as fp32 MFMA (default) / bf16 x 3 split
    else if (variant == 230 || variant == 231) t.head_sums = variant - 230;           // predictor head from a normalise pass / from the last GEMM's epilogue sums (default)
    else if (variant == 220 || variant == 221) t.gemm_persist = variant - 220;         // multi-round bf16 pointwise launches one tile per workgroup / on the persistent kernel (default)
    else if (variant >= 200 && variant <= 202) t.slab_xcd_remap = variant - 200;       // slab kernel tile order plain / XCD-contiguous (default) / + column pairs per XCD for wide weight panels
    else if (variant >= 0 && variant <= 7) t.gemm_variant = variant;                    // kernel family / forced tile height of the forward GEMM launcher (gemm_mfma.hip: launch_gemm)
    else ok = 0;
    if (!ok) return FS2_ERR_ARG;
    ++t.gen;
    return FS2_OK;
}
Tuning& op_tuning() {
    static thread_local Tuning t;
    return t;
}
}  // namespace fs2

extern "C" {

int fs2_op_convert(int32_t sdt, int32_t ddt, const void* src, void* dst, size_t n, void* stream) {
    ConvertArgs a{src, dst, n};
    return launch_convert(a, sdt, ddt, (hipStream_t)stream);
}

int fs2_op_predictor(int32_t dtype, const void* x, const void* w, const float* bias, const float* ln_g,
                     const float* ln_b, const float* head_w, float head_b, const uint8_t* mask, float* pred,
                     void* packed_scratch, int32_t B, int32_t S, int32_t H, int32_t nlayers, int32_t taps,
                     void* stream) {
    if (!predictor_fused_supported(dtype, H, taps, nlayers, S)) return FS2_ERR_SHAPE;
    if (!x || !w || !bias || !ln_g || !ln_b || !head_w || !pred || !packed_scratch) return FS2_ERR_ARG;
    const size_t lb = predictor_packed_bytes_per_layer();
    for (int l = 0; l < nlayers; ++l) {
        const int r = launch_pack_predictor_weights((const char*)w + (size_t)l * H * taps * H * 2,
                                                    (char*)packed_scratch + lb * l, (hipStream_t)stream);
        if (r != FS2_OK) return r;
    }
    PredictorArgs a;
    a.x = x; a.wpk = packed_scratch; a.bias = bias; a.ln_g = ln_g; a.ln_b = ln_b; a.head_w = head_w; a.head_b = head_b;
    a.mask = mask; a.pred = pred; a.B = B; a.S = S; a.H = H; a.nlayers = nlayers; a.taps = taps; a.eps = 1e-5f;
    return launch_predictor_fused(a, (hipStream_t)stream);
}

int fs2_op_predictor_dw(int32_t dtype, const void* x, const float* dw_w, const float* dw_b, const void* w, const float* bias,
                        const float* ln_g, const float* ln_b, const float* head_w, float head_b, const uint8_t* mask, float* pred,
                        void* packed_scratch, int32_t B, int32_t S, int32_t H, int32_t nlayers, void* stream) {
    if (!predictor_fused_supported(dtype, H, 3, nlayers, S)) return FS2_ERR_SHAPE;
    if (!x || !dw_w || !dw_b || !w || !bias || !ln_g || !ln_b || !head_w || !pred || !packed_scratch) return FS2_ERR_ARG;
    const size_t lb = predictor_packed_bytes_per_layer(1);
    for (int l = 0; l < nlayers; ++l) {
        const int r = launch_pack_predictor_weights((const char*)w + (size_t)l * H * H * 2, (char*)packed_scratch + lb * l, (hipStream_t)stream, 1);
        if (r != FS2_OK) return r;
    }
    PredictorArgs a;
    a.x = x; a.wpk = packed_scratch; a.dw_w = dw_w; a.dw_b = dw_b; a.bias = bias; a.ln_g = ln_g; a.ln_b = ln_b; a.head_w = head_w; a.head_b = head_b;
    a.mask = mask; a.pred = pred; a.B = B; a.S = S; a.H = H; a.nlayers = nlayers; a.taps = 3; a.eps = 1e-5f;
    return launch_predictor_fused(a, (hipStream_t)stream);
}

int fs2_op_soft_dtw(const float* x, const float* y, int32_t B, int32_t N, int32_t M, int32_t D, float gamma, float* out,
                    void* stream) {
    if (!x || !y || !out) return FS2_ERR_ARG;
    SoftDtwArgs a{x, y, out, B, N, M, D, gamma};
    return launch_soft_dtw(a, (hipStream_t)stream);
}
size_t fs2_op_soft_dtw_grad_scratch_bytes(int32_t B, int32_t N, int32_t M) { return soft_dtw_grad_scratch_bytes(B, N, M); }
int fs2_op_soft_dtw_grad(const float* x, const float* y, int32_t B, int32_t N, int32_t M, int32_t D, float gamma, float* out,
                         float* grad_x, void* scratch, size_t scratch_bytes, void* stream) {
    if (!x || !y || !out) return FS2_ERR_ARG;
    SoftDtwArgs a{x, y, out, B, N, M, D, gamma};
    return launch_soft_dtw_grad(a, grad_x, scratch, scratch_bytes, (hipStream_t)stream);
}

size_t fs2_op_masked_loss_ws_bytes(void) { return fs2::masked_loss_ws_bytes(); }

int fs2_op_masked_loss(const float* pred, const void* truth, int32_t truth_kind, const uint8_t* pad_mask, int64_t rows,
                       int32_t inner, int32_t kind, void* ws, float* out2, void* stream) {
    if (!pred || !truth || !pad_mask || !ws || !out2) return FS2_ERR_ARG;
    if ((kind != 0 && kind != 1) || (truth_kind != 0 && truth_kind != 1)) return FS2_ERR_ARG;
    fs2::LossArgs a;
    a.pred = pred; a.truth = truth; a.mask = pad_mask; a.ws = ws; a.out = out2;
    a.rows = rows; a.inner = inner; a.kind = kind; a.truth_kind = truth_kind;
    return fs2::launch_masked_loss(a, (hipStream_t)stream);
}

int fs2_op_set_vocoder_fused_resblock(int32_t on) {
    fs2::op_tuning().voc_fused_resblock = on;
    return FS2_OK;
}

int fs2_op_set_vocoder_lds_limit(int32_t kib) {
    fs2::op_tuning().voc_lds_limit = kib;
    return FS2_OK;
}



int fs2_op_set_gemm_variant(int32_t variant) { return fs2::apply_knob(fs2::op_tuning(), variant); }

int fs2_op_gemm(int32_t dtype, int32_t out_dtype, const void* x, const void* w, const float* bias, void* c,
                int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S, int32_t relu, void* stream) {
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = c;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = relu;
    return launch_gemm(a, dtype, out_dtype, (hipStream_t)stream);
}

int fs2_op_gemm_relu_dropout(int32_t dtype, const void* x, const void* w, const float* bias, void* c, int32_t M, int32_t N, int32_t Cin,
                             int32_t taps, int32_t S, float p, uint64_t seed, uint64_t key, void* stream) {
    if (!(p > 0.f && p < 1.f)) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = c;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = 1;
    a.drop_p = p; a.drop_seed = seed; a.drop_key = key;
    return launch_gemm(a, dtype, dtype, (hipStream_t)stream);
}

int fs2_op_gemm_add(int32_t dtype, const void* x, const void* w, const float* bias, const void* addend, void* c, int32_t M, int32_t N,
                    int32_t Cin, int32_t taps, int32_t S, void* stream) {
    if (!addend) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = c;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = 0;
    a.epi_res = addend;  // the slab kernel's residual epilogue without statistics: the accumulators start AT the addend
    if (N < 192 || M % S || !(taps & 1)) return FS2_ERR_SHAPE;  // slab kernel only
    return launch_gemm(a, dtype, dtype, (hipStream_t)stream);
}

int fs2_op_gemm_rowscale(const void* x, const void* w, const float* bias, const float* rowstats, const float* wg,
                         void* c, int32_t M, int32_t N, int32_t Cin, void* stream) {
    if (!x || !w || !bias || !rowstats || !wg || !c) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = c;
    a.M = M; a.N = N; a.K = Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = 1; a.pad = 0; a.S = M; a.relu = 0;
    a.rs_stats = rowstats; a.rs_wg = wg;
    return launch_gemm(a, FS2_BF16, FS2_BF16, (hipStream_t)stream);
}
int fs2_op_rowstats_finish(const float* parts, int32_t nparts, int32_t ncols, float eps, float* out, int32_t M, void* stream) {
    return launch_rowstats_finish(parts, nparts, ncols, eps, out, M, (hipStream_t)stream);
}
int fs2_op_gemm_head(const void* x, const void* w, const float* bias, const float* head_gw, float* stats_out, float* head_out, int32_t M,
                     int32_t N, int32_t Cin, int32_t relu, void* stream) {
    if (!x || !w || !bias || !head_gw || !stats_out || !head_out) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = nullptr;
    a.M = M; a.N = N; a.K = Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = 1; a.pad = 0; a.S = M; a.relu = relu;
    a.stats_out = stats_out; a.head_gw = head_gw; a.head_out = head_out; a.ln_eps = 1e-5f;
    return launch_gemm(a, FS2_BF16, FS2_BF16, (hipStream_t)stream);
}
int fs2_op_head_finish(const float* parts, const float* dots, int32_t nparts, int32_t ncols, float eps, float sum_gw, float cst,
                       const uint8_t* mask, float* pred, int32_t M, void* stream) {
    return launch_head_finish(parts, dots, nparts, ncols, eps, sum_gw, cst, mask, pred, M, (hipStream_t)stream);
}

int fs2_op_gemm_splitk_choice(int32_t dtype, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S) {
    return gemm_splitk_choice(M, N, Cin, taps, S, dtype);
}

int fs2_op_gemm_splitk(int32_t dtype, int32_t out_dtype, const void* x, const void* w, void* c, float* part, int32_t M, int32_t N,
                       int32_t Cin, int32_t taps, int32_t S, int32_t ksplit, int32_t accumulate, void* stream) {
    if (ksplit < 2 || !part) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = nullptr; a.C = part;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = 0;
    a.ksplit = ksplit;
    const int r = launch_gemm(a, dtype, FS2_F32, (hipStream_t)stream);
    if (r != FS2_OK) return r;
    return launch_split_k_reduce(part, c, (size_t)M * N, ksplit, accumulate, out_dtype, (hipStream_t)stream);
}

int fs2_op_gemm_gated(int32_t dtype, const void* x, const void* w, const float* bias, const void* gate, float scale, void* c, int32_t M,
                      int32_t N, int32_t Cin, int32_t taps, int32_t S, void* stream) {
    if (!gate) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = c;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = 0;
    a.gate = gate; a.gate_scale = scale;
    return launch_gemm(a, dtype, dtype, (hipStream_t)stream);
}

int fs2_op_gemm_ln(int32_t dtype, const void* x, const void* w, const float* bias, const void* res,
                   const float* ln_g, const float* ln_b, const float* dot_w, float dot_b, const uint8_t* mask,
                   float* pred, void* y, void* tmp, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S,
                   int32_t relu, void* stream) {
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = y;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = relu;
    a.res = res; a.ln_g = ln_g; a.ln_b = ln_b; a.dot_w = dot_w; a.dot_b = dot_b; a.mask = mask; a.pred = pred;
    a.ln_tmp = tmp;
    return launch_gemm(a, dtype, dtype, (hipStream_t)stream);
}

int fs2_op_gemm_ln_tape(int32_t dtype, const void* x, const void* w, const float* bias, const void* res, const float* ln_g,
                        const float* ln_b, void* y, void* z_out, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S,
                        int32_t relu, void* stream) {
    if (!x || !w || !ln_g || !ln_b || !y || !z_out) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = y;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = relu;
    a.res = res; a.ln_g = ln_g; a.ln_b = ln_b; a.z_out = z_out;
    return launch_gemm(a, dtype, dtype, (hipStream_t)stream);
}

int fs2_op_gemm_ln_tape_dropout(int32_t dtype, const void* x, const void* w, const float* bias, const void* res, const float* ln_g,
                                const float* ln_b, void* y, void* z_out, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S,
                                int32_t relu, float p, uint64_t seed, uint64_t key, void* stream) {
    if (!x || !w || !ln_g || !ln_b || !y || !z_out) return FS2_ERR_ARG;
    if (!(p >= 0.f && p < 1.f)) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = y;
    a.M = M; a.N = N; a.K = taps * Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = taps; a.pad = (taps - 1) / 2; a.S = S; a.relu = relu;
    a.res = res; a.ln_g = ln_g; a.ln_b = ln_b; a.z_out = z_out;
    a.drop_p = p; a.drop_seed = seed; a.drop_key = key;
    return launch_gemm(a, dtype, dtype, (hipStream_t)stream);
}

size_t fs2_op_attention_scratch_bytes(int32_t dtype, int32_t B, int32_t S, int32_t H, int32_t heads, size_t* bits_bytes) {
    (void)heads;
    const size_t Spad = ((size_t)S + 63) / 64 * 64;
    if (bits_bytes) *bits_bytes = (size_t)B * (Spad / 64) * 8;
    return (size_t)B * H * Spad * (dtype == FS2_BF16 ? 2 : 4);
}

int fs2_op_attention(int32_t dtype, const void* qkv, const uint8_t* key_pad_mask, void* out, void* vt_scratch,
                     uint64_t* bits_scratch, int32_t B, int32_t S, int32_t H, int32_t heads, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Spad = (S + 63) / 64 * 64;
    MaskBitsArgs mb{key_pad_mask, bits_scratch, B, S, Spad / 64};
    int r = launch_mask_bits(mb, st);
    if (r != FS2_OK) return r;
    AttnArgs a;
    a.qkv = qkv; a.vt = vt_scratch; a.kbits = bits_scratch; a.out = out;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.Spad = Spad; a.nw64 = Spad / 64;
    a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)(H / heads)));
    r = launch_transpose_v(a, dtype, st);
    if (r != FS2_OK) return r;
    return launch_attention(a, dtype, st);
}

// Encoder-side fused launch (attention.hip attn_out_ln_kernel, r06): out = LayerNorm(res + MHA-core(qkv) w_out^T + bias), bf16, H = 256, 2 heads.
// scratch: H * H * 2 bytes (w_out in fragment order) + B * ceil(S / 64) * 8 bytes (valid-key words), 16-byte aligned.
int fs2_op_attn_out_ln(int32_t dtype, const void* qkv, const uint8_t* key_pad_mask, const void* w_out, const float* bias, const void* res,
                       const float* ln_g, const float* ln_b, void* out, void* scratch, int32_t B, int32_t S, int32_t H, int32_t heads, void* stream) {
    if (!qkv || !key_pad_mask || !w_out || !bias || !res || !ln_g || !ln_b || !out || !scratch || B <= 0) return FS2_ERR_ARG;
    if (!attn_out_ln_supported(dtype, H, heads, S)) return FS2_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nw64 = (S + 63) / 64;
    uint64_t* bits = (uint64_t*)((char*)scratch + (size_t)H * H * 2);
    MaskBitsArgs mb{key_pad_mask, bits, B, S, nw64};
    int r = launch_mask_bits(mb, st);
    if (r != FS2_OK) return r;
    r = launch_pack_predictor_weights(w_out, scratch, st, 1);
    if (r != FS2_OK) return r;
    AttnOutArgs a;
    a.qkv = qkv; a.kbits = bits; a.wpk = scratch; a.bias = bias; a.res = res; a.ln_g = ln_g; a.ln_b = ln_b; a.out = out;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.nw64 = nw64;
    a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)(H / heads))); a.eps = 1e-5f;
    return launch_attn_out_ln(a, st);
}

// The split-arithmetic attention on an fp32 (B*S, 3H) qkv tensor: head / tail split pass (the engine's in-projection writes the two
// halves itself), then attention_kernel<bf16, .., X3>; out = fp32 (B*S, H).  split_scratch: 2 * B*S*3H bf16.
int fs2_op_attention_x3(const float* qkv, const uint8_t* key_pad_mask, float* out, void* split_scratch, uint64_t* bits_scratch,
                        int32_t B, int32_t S, int32_t H, int32_t heads, void* stream) {
    if (!qkv || !key_pad_mask || !out || !split_scratch || !bits_scratch || B <= 0 || S <= 0 || H <= 0 || heads <= 0 || H % heads) return FS2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int Spad = (S + 63) / 64 * 64;
    MaskBitsArgs mb{key_pad_mask, bits_scratch, B, S, Spad / 64};
    int r = launch_mask_bits(mb, st);
    if (r != FS2_OK) return r;
    const size_t n = (size_t)B * S * 3 * H;
    void* lo = (char*)split_scratch + n * 2;
    r = launch_split_hi_lo(qkv, split_scratch, lo, n, st);
    if (r != FS2_OK) return r;
    AttnArgs a;
    a.qkv = split_scratch; a.qkv_lo = lo; a.vt = nullptr; a.kbits = bits_scratch; a.out = out;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.Spad = Spad; a.nw64 = Spad / 64;
    a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)(H / heads)));
    return launch_attention(a, FS2_F32, st);
}

// fp32 GEMM (bf16 x 3 split products when split != 0) whose result leaves as two bf16 tensors, head + tail
int fs2_op_gemm_split_out(const void* x, const void* w, const float* bias, void* c_hi, void* c_lo, int32_t M, int32_t N, int32_t Cin,
                          int32_t split, void* stream) {
    if (!x || !w || !c_hi || !c_lo) return FS2_ERR_ARG;
    GemmArgs a;
    a.X = x; a.W = w; a.bias = bias; a.C = c_hi; a.C_lo = c_lo;
    a.M = M; a.N = N; a.K = Cin; a.ldx = Cin; a.ldc = N;
    a.Cin = Cin; a.taps = 1; a.pad = 0; a.S = M; a.relu = 0; a.split = split != 0;
    return launch_gemm(a, FS2_F32, FS2_F32, (hipStream_t)stream);
}

int fs2_op_attention_train(int32_t dtype, const void* qkv, const uint8_t* key_pad_mask, void* out, void* vt_scratch,
                           uint64_t* bits_scratch, float* lse2, int32_t B, int32_t S, int32_t H, int32_t heads, float drop_p,
                           uint64_t drop_seed, uint64_t drop_key, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Spad = (S + 63) / 64 * 64;
    MaskBitsArgs mb{key_pad_mask, bits_scratch, B, S, Spad / 64};
    int r = launch_mask_bits(mb, st);
    if (r != FS2_OK) return r;
    AttnArgs a;
    a.qkv = qkv; a.vt = vt_scratch; a.kbits = bits_scratch; a.out = out;
    a.B = B; a.S = S; a.H = H; a.heads = heads; a.Spad = Spad; a.nw64 = Spad / 64;
    a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)(H / heads)));
    a.lse2 = lse2; a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_key = drop_key;
    r = launch_transpose_v(a, dtype, st);
    if (r != FS2_OK) return r;
    return launch_attention(a, dtype, st);
}
int32_t fs2_op_attention_bwd_supported(int32_t dtype, int32_t H, int32_t heads) { return attention_bwd_supported(dtype, H, heads) ? 1 : 0; }
int fs2_op_attention_bwd(int32_t dtype, const void* qkv, const void* dout, const float* lse2, const float* delta,
                         const uint8_t* key_pad_mask, void* dqkv, int32_t B, int32_t S, int32_t H, int32_t heads, float drop_p,
                         uint64_t drop_seed, uint64_t drop_key, void* stream) {
    AttnBwdArgs a{qkv, dout, lse2, delta, key_pad_mask, dqkv, B, S, H, heads,
                  (float)(1.4426950408889634 / sqrt((double)(H / heads))), (float)(1.0 / sqrt((double)(H / heads))), drop_p, drop_seed, drop_key};
    return launch_attention_bwd(a, dtype, (hipStream_t)stream);
}

int fs2_op_layernorm(int32_t dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                     const float* dot_w, float dot_b, const uint8_t* mask, float* pred, int32_t M, int32_t H,
                     void* stream) {
    LayerNormArgs a;
    a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y;
    a.dot_w = dot_w; a.dot_b = dot_b; a.mask = mask; a.pred = pred;
    a.M = M; a.H = H; a.eps = 1e-5f;
    return launch_layernorm(a, dtype, (hipStream_t)stream);
}
int fs2_op_layernorm_head(int32_t dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                          const float* dot_w, const float* dot_b_dev, const uint8_t* mask, float* pred, int32_t M, int32_t H,
                          void* stream) {
    if (!dot_w || !dot_b_dev || !pred) return FS2_ERR_ARG;
    LayerNormArgs a;
    a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y;
    a.dot_w = dot_w; a.dot_b = 0.f; a.mask = mask; a.pred = pred;
    a.M = M; a.H = H; a.eps = 1e-5f;
    a.dot_b_dev = dot_b_dev;
    return launch_layernorm(a, dtype, (hipStream_t)stream);
}
int fs2_op_layernorm_dropout(int32_t dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                             int32_t M, int32_t H, float drop_p, uint64_t seed, uint64_t key, void* stream) {
    if (!(drop_p >= 0.f && drop_p < 1.f) || !y) return FS2_ERR_ARG;
    LayerNormArgs a;
    a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y;
    a.dot_w = nullptr; a.dot_b = 0.f; a.mask = nullptr; a.pred = nullptr;
    a.M = M; a.H = H; a.eps = 1e-5f;
    a.drop_p = drop_p; a.drop_seed = seed; a.drop_key = key;
    return launch_layernorm(a, dtype, (hipStream_t)stream);
}

int fs2_op_dwconv(int32_t dtype, const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t S,
                  int32_t C, int32_t k, void* stream) {
    DwConvArgs a{x, w, bias, y, B, S, C, k, (k - 1) / 2};
    return launch_dwconv(a, dtype, (hipStream_t)stream);
}

int fs2_op_durations(const float* dur_pred, const uint8_t* src_mask, const int32_t* forced, int32_t* dur,
                     int32_t* cum, int32_t* totals, int32_t* guard, int32_t B, int32_t L, void* stream) {
    DurationArgs a{dur_pred, src_mask, forced, dur, cum, totals, guard, B, L};
    return launch_durations(a, (hipStream_t)stream);
}

int fs2_op_regulate(int32_t dtype, const void* x, const int32_t* cum, const int32_t* totals, void* y,
                    uint8_t* tgt_mask, int32_t B, int32_t L, int32_t T, int32_t H, void* stream) {
    RegulateArgs a{x, cum, totals, y, tgt_mask, B, L, T, H};
    return launch_regulate(a, dtype, (hipStream_t)stream);
}

int fs2_op_bucket_embed(int32_t dtype, const void* x, const float* pred, const float* bins, const float* emb,
                        int32_t nbins, float std, float mean, const float* pe, const float* spk, void* y,
                        int32_t* idx_out, int32_t B, int32_t T, int32_t H, void* stream) {
    BucketArgs a{x, pred, bins, emb, nbins, std, mean, pe, spk, y, idx_out, B, T, H, nullptr};
    return launch_bucket_embed(a, dtype, (hipStream_t)stream);
}

int fs2_op_embed(int32_t dtype, const int64_t* phones, const float* table, const float* pe, const float* spk,
                 void* x, uint8_t* src_mask, int32_t B, int32_t L, int32_t H, int32_t n_phones, void* stream) {
    EmbedArgs a{phones, table, pe, spk, x, src_mask, B, L, H, n_phones};
    return launch_embed(a, dtype, (hipStream_t)stream);
}

int fs2_op_spk_proj(const float* dvec, const float* w, const float* b, float* spk, int32_t B, int32_t H,
                    int32_t Din, void* stream) {
    SpkProjArgs a{dvec, w, b, spk, B, H, Din};
    return launch_spk_proj(a, (hipStream_t)stream);
}

// ---- training step (f4): backward operators -------------------------------------------------------------------------
static BGemmArgs bgemm_args(const fs2_bgemm_desc* d, const void* A, const void* B, void* C, const float* bias, float* ws) {
    BGemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.ws = ws;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.sAm = d->sAm; a.sAk = d->sAk; a.sBk = d->sBk; a.sBn = d->sBn; a.ldc = d->ldc;
    a.nb1 = d->nb1 > 0 ? d->nb1 : 1; a.nb2 = d->nb2 > 0 ? d->nb2 : 1;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.alpha = d->alpha; a.beta = d->beta; a.splitk = d->splitk > 1 ? d->splitk : 1;
    a.seg = d->seg; a.taps = d->taps > 1 ? d->taps : 1; a.Kin = d->Kin;
    a.a_shift0 = d->a_shift0; a.a_shift_step = d->a_shift_step; a.sBtap = d->sBtap;
    a.b_shift0 = d->b_shift0; a.b_shift_step = d->b_shift_step; a.c_dtype = d->c_dtype;
    return a;
}
size_t fs2_op_bgemm_ws_bytes(const fs2_bgemm_desc* d) {
    return d ? bgemm_ws_bytes(bgemm_args(d, nullptr, nullptr, nullptr, nullptr, nullptr)) : 0;
}
int32_t fs2_op_bgemm_tn256(const fs2_bgemm_desc* d) {
    return d && bgemm_tn256_eligible(bgemm_args(d, nullptr, nullptr, nullptr, nullptr, nullptr)) ? 1 : 0;
}
int fs2_op_bgemm(int32_t dtype, const fs2_bgemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                 float* ws, void* stream) {
    if (!d || !A || !B || !C) return FS2_ERR_ARG;
    return launch_bgemm(bgemm_args(d, A, B, C, bias, ws), dtype, (hipStream_t)stream);
}
int fs2_op_bgemm_softmax_bwd(int32_t dtype, const fs2_bgemm_desc* d, const void* A, const void* B, void* C, const void* P,
                             const float* delta, float drop_p, uint64_t drop_seed, uint64_t drop_key, void* stream) {
    if (!d || !A || !B || !C || !P || !delta) return FS2_ERR_ARG;
    BGemmArgs a = bgemm_args(d, A, B, C, nullptr, nullptr);
    a.epi_p = P; a.epi_delta = delta; a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_key = drop_key;
    return launch_bgemm(a, dtype, (hipStream_t)stream);
}
int fs2_op_attn_delta(int32_t dtype, const void* dout, const void* out, float* delta, int32_t B, int32_t S, int32_t H,
                      int32_t heads, void* stream) {
    AttnDeltaArgs a{dout, out, delta, B, S, H, heads};
    return launch_attn_delta(a, dtype, (hipStream_t)stream);
}
int32_t fs2_op_layernorm_bwd_parts(int32_t M) { return layernorm_bwd_parts(M); }
int fs2_op_layernorm_bwd(int32_t dtype, const void* z, const void* res, const void* dy, const float* gamma, void* dz,
                         float* part, int32_t M, int32_t H, int32_t relu_mask, void* stream) {
    LayerNormBwdArgs a{z, res, dy, gamma, dz, part, M, H, layernorm_bwd_parts(M), 1e-5f, relu_mask};
    return launch_layernorm_bwd(a, dtype, (hipStream_t)stream);
}
int fs2_op_layernorm_bwd_dropout(int32_t dtype, const void* z, const void* res, const void* dy, const float* gamma, void* dz,
                                 float* part, int32_t M, int32_t H, int32_t relu_mask, float drop_p, uint64_t seed, uint64_t key,
                                 void* stream) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) return FS2_ERR_ARG;
    LayerNormBwdArgs a{z, res, dy, gamma, dz, part, M, H, layernorm_bwd_parts(M), 1e-5f, relu_mask};
    a.drop_p = drop_p; a.drop_seed = seed; a.drop_key = key;
    return launch_layernorm_bwd(a, dtype, (hipStream_t)stream);
}
int fs2_op_layernorm_bwd_masked(int32_t dtype, const void* z, const void* res, const void* dy, const float* gamma, void* dz, void* dzm,
                                float* part, int32_t M, int32_t H, int32_t relu_mask, float out_p, uint64_t seed, uint64_t out_key,
                                void* stream) {
    if (!dzm || !(out_p > 0.f && out_p < 1.f)) return FS2_ERR_ARG;
    LayerNormBwdArgs a{z, res, dy, gamma, dz, part, M, H, layernorm_bwd_parts(M), 1e-5f, relu_mask};
    a.drop_seed = seed; a.dzm = dzm; a.out_p = out_p; a.out_key = out_key;
    return launch_layernorm_bwd(a, dtype, (hipStream_t)stream);
}
size_t fs2_op_col_sum_ws_bytes(int32_t M, int32_t N, int32_t seg) { return col_sum_ws_bytes(M, N, seg); }
int fs2_op_col_sum(int32_t dtype, const void* x, float* out, float* ws, int32_t M, int32_t N, int32_t ldx, int32_t seg,
                   int32_t accumulate, float scale, void* stream) {
    ColSumArgs a{x, out, ws, M, N, ldx, seg, accumulate, scale};
    return launch_col_sum(a, dtype, (hipStream_t)stream);
}
int fs2_op_col_sum2(int32_t dtype, const void* x, float* out, float* out2, int32_t n1, float* ws, int32_t M, int32_t N, int32_t ldx,
                    int32_t accumulate, int32_t accumulate2, float scale, void* stream) {
    ColSumArgs a{x, out, ws, M, N, ldx, 0, accumulate, scale};
    a.out2 = out2; a.n1 = n1; a.accumulate2 = accumulate2;
    if (!out2) return FS2_ERR_ARG;
    return launch_col_sum(a, dtype, (hipStream_t)stream);
}
int fs2_op_col_sum_weighted(int32_t dtype, const void* x, const float* row_w, float* out, float* ws, int32_t M, int32_t N, int32_t ldx,
                            int32_t accumulate, float scale, void* stream) {
    if (!row_w) return FS2_ERR_ARG;
    ColSumArgs a{x, out, ws, M, N, ldx, 0, accumulate, scale};
    a.row_w = row_w;
    return launch_col_sum(a, dtype, (hipStream_t)stream);
}
int fs2_op_softmax_fwd(int32_t dtype, const float* s, const uint8_t* key_pad, void* p, int32_t B, int32_t heads, int32_t S,
                       float scale, void* stream) {
    SoftmaxArgs a{s, nullptr, p, key_pad, B, heads, S, scale};
    return launch_softmax_fwd(a, dtype, (hipStream_t)stream);
}
int fs2_op_softmax_bwd(int32_t dtype, const float* dp, const void* p, void* ds, int32_t B, int32_t heads, int32_t S,
                       float scale, void* stream) {
    SoftmaxArgs a{dp, p, ds, nullptr, B, heads, S, scale};
    return launch_softmax_bwd(a, dtype, (hipStream_t)stream);
}
int fs2_op_ew(int32_t dtype, int32_t op, const void* a_, const void* b, void* out, size_t n, float alpha, float beta,
              void* stream) {
    EwArgs a{a_, b, out, n, alpha, beta, op};
    return launch_ew(a, dtype, (hipStream_t)stream);
}
size_t fs2_op_scatter_rows_ws_bytes(int32_t R, int32_t H, int32_t V) { return scatter_rows_ws_bytes(R, H, V); }
int fs2_op_scatter_rows(int32_t dtype, const void* x, const int32_t* idx32, const int64_t* idx64, float* table, float* ws, int32_t R,
                        int32_t H, int32_t V, int32_t skip_row, void* stream) {
    ScatterRowsArgs a{x, idx32, idx64, table, R, H, V, skip_row, ws};
    return launch_scatter_rows(a, dtype, (hipStream_t)stream);
}
int fs2_op_regulate_bwd(int32_t dtype, const void* dy, const int32_t* cum, void* dx, int32_t B, int32_t L, int32_t T,
                        int32_t H, void* stream) {
    RegulateBwdArgs a{dy, cum, dx, B, L, T, H};
    return launch_regulate_bwd(a, dtype, (hipStream_t)stream);
}
int fs2_op_masked_loss_bwd(const float* pred, const void* truth, int32_t truth_kind, const uint8_t* pad_mask,
                           const float* stat, float* dpred, int64_t rows, int32_t inner, int32_t kind, float alpha,
                           void* stream) {
    LossBwdArgs a{pred, truth, pad_mask, stat, dpred, rows, inner, kind, truth_kind, alpha};
    return launch_masked_loss_bwd(a, (hipStream_t)stream);
}
int fs2_op_bucket_embed_utt(int32_t dtype, const void* x, const float* values, const float* bins, const float* emb, int32_t nbins,
                            void* y, int32_t* idx_out, int32_t B, int32_t T, int32_t H, void* stream) {
    BucketArgs a{x, values, bins, emb, nbins, 1.f, 0.f, nullptr, nullptr, y, idx_out, B, T, H, nullptr};
    a.pred_per_utt = 1;
    return launch_bucket_embed(a, dtype, (hipStream_t)stream);
}
int fs2_op_dropout(int32_t dtype, const void* x, void* y, size_t n, float prob, uint64_t seed, uint64_t key, void* stream) {
    DropoutArgs a{x, y, n, prob, seed, key};
    return launch_dropout(a, dtype, (hipStream_t)stream);
}
int fs2_op_row_dot(int32_t dtype, const void* y, const float* w, const float* b, const uint8_t* mask, float* pred, int64_t M,
                   int32_t H, void* stream) {
    RowDotArgs a{y, w, b, mask, pred, (long)M, H};
    return launch_row_dot(a, dtype, (hipStream_t)stream);
}
int fs2_op_dwconv_dgrad(int32_t dtype, const void* dy, const float* w, void* dx, int32_t B, int32_t S, int32_t C, int32_t k,
                        void* stream) {
    DwConvArgs a{dy, w, nullptr, dx, B, S, C, k, (k - 1) / 2};
    a.flip = 1;
    return launch_dwconv(a, dtype, (hipStream_t)stream);
}
int32_t fs2_op_dwconv_wgrad_parts(int32_t B, int32_t S) { return dwconv_wgrad_parts(B, S); }
int fs2_op_dwconv_wgrad(int32_t dtype, const void* dy, const void* x, float* part, int32_t B, int32_t S, int32_t C, int32_t k,
                        void* stream) {
    DwConvWgradArgs a{dy, x, part, B, S, C, k, (k - 1) / 2};
    return launch_dwconv_wgrad(a, dtype, (hipStream_t)stream);
}
int fs2_op_fold_conv2(int32_t wf_dtype, const float* G, const float* bg, const float* W21, const float* b21, void* Wf, float* bf,
                      int32_t H, int32_t F, void* stream) {
    FoldConv2Args a{G, bg, W21, b21, Wf, bf, H, F};
    return launch_fold_conv2(a, wf_dtype, (hipStream_t)stream);
}
int fs2_op_unfold_conv2(const float* dWf, const float* dbf, const float* G, const float* bg, const float* W21, float* dG,
                        float* dbg, float* dW21, float* db21, int32_t H, int32_t F, void* stream) {
    UnfoldConv2Args a{dWf, dbf, G, bg, W21, dG, dbg, dW21, db21, H, F};
    return launch_unfold_conv2(a, (hipStream_t)stream);
}
int fs2_op_transpose_weight(int32_t dtype, const void* src, void* dst, int32_t N, int32_t Cin, int32_t taps, void* stream) {
    TransposeWeightArgs a{src, dst, N, Cin, taps};
    return launch_transpose_weight(a, dtype, (hipStream_t)stream);
}
size_t fs2_op_sum_sq_ws_bytes(size_t n) { return sum_sq_ws_bytes(n); }
int fs2_op_sum_sq(const float* x, size_t n, float* ws, float* out, void* stream) {
    return launch_sum_sq(x, n, ws, out, (hipStream_t)stream);
}
int fs2_op_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale, void* stream) {
    AdamWArgs a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq, max_norm, grad_scale};
    return launch_adamw(a, (hipStream_t)stream);
}
int fs2_op_adamw_shadow(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale,
                        void* stream) {
    AdamWArgs a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq, max_norm, grad_scale, shadow_bf16};
    return launch_adamw(a, (hipStream_t)stream);
}
int64_t fs2_op_transpose_weight_tiles(int32_t N, int32_t Cin, int32_t taps) {
    return (int64_t)((N + 63) / 64) * ((Cin + 63) / 64) * taps;
}
int fs2_op_transpose_weight_batch(const int64_t* table_dev, int32_t n, int64_t tiles, void* stream) {
    return launch_transpose_weight_batch((const long long*)table_dev, n, tiles, (hipStream_t)stream);
}
int fs2_op_bucket_embed_target(int32_t dtype, const void* x, const float* target, const float* bins, const float* emb,
                               int32_t nbins, float std, float mean, const float* pe, const float* spk, void* y,
                               int32_t* idx_out, int32_t B, int32_t T, int32_t H, void* stream) {
    BucketArgs a{x, target, bins, emb, nbins, std, mean, pe, spk, y, idx_out, B, T, H, nullptr};
    a.bucket_src = target;
    return launch_bucket_embed(a, dtype, (hipStream_t)stream);
}

}  // extern "C"
