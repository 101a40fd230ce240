// Internal launcher interface between the engine (host C++) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fs2.h"

namespace fs2 {

// A/B switches of the launchers (kernel family, tile order, which fused form a launch takes).  NOT process state: an engine owns one
// (fs2_set_tuning, copied by fs2_clone), the operator-level entry points (fs2_op_*: tests, the training step) use the CALLING
// THREAD's (fs2_op_set_gemm_variant -> op_tuning()).  Every launcher reads the one its arguments point at (Args::tune; null = the
// calling thread's).  Defaults are what measured fastest; every non-default value is kept for A/B or for the tests that pin two
// forms against each other.  (Until r05 these were ~20 process-wide ints read unsynchronised by every engine and pipeline thread.)
struct Tuning {
    int gemm_variant = 0;      // 0 auto; 1 128x128 register-staged; 2 128x256 DMA ring; 3/4/5 slab kernel 128/192/256-row tiles; 6/7 32/64-row
    int gemm_wres = 1;         // bf16 K = 256 plain GEMMs on the weight-resident kernel: 0 never, 1 where it pays, 2 wherever it applies
    int head_sums = 1;         // wide predictors: last LayerNorm + Linear head from the last GEMM's epilogue sums (GemmArgs::head_out) / a normalise pass
    int gemm_persist = 1;      // bf16 pointwise launches of more tiles than CUs on the persistent kernel (gemm_persist.hip)
    int slab_xcd_remap = 1;    // slab kernel tile order: 0 plain, 1 XCD-contiguous (default), 2 = 1 + column pairs per XCD where the weight panel exceeds an L2 (fewer bytes fetched, 1 % slower: r05)
    int split_f32 = 0;         // operator level only (tests): every fp32 slab launch in the bf16 x 3 split arithmetic
    int attn_pipe = 3;         // 0 attention_kernel only; 1 / 2 / 4 the pipelined kernel with 32 / 64 / 96 queries per wave; 3 by size
    int attn_x3 = 1;           // fp32-storage split modes: attention on bf16 x 3 split products (1) or fp32 MFMA (0)
    int enc_attn_out = 1;      // engine: the encoder's attention + out-projection + residual + LayerNorm as one launch (attn_out_ln_kernel; bf16, H = 256, 2 heads)
    int pred_fuse_embed = 1;   // engine: the variance encoder's bucketize + embedding add as the tail of its predictor launch
    int attn_bwd_nb = 1, attn_bwd_nb_dq = 0;  // attention backward: 16-row blocks per wave of the (dK, dV) / dQ launch (dQ: 0 = by size)
    int colsum_fused = 0;      // column sums in one launch (device-scope fences: slower) / two
    int bgemm_full = 1, bgemm_xcd = 1, bgemm_tn256 = 1;
    int voc_lds_limit = 0;     // KiB cap on a vocoder conv workgroup's slab; 0 = heuristic
    int voc_fused_resblock = 1;
    unsigned long long gen = 0;  // bumped by every accepted change: part of an engine's hipGraph keys
};
Tuning& op_tuning();                       // the calling thread's (capi_ops.hip)
int apply_knob(Tuning& t, int variant);    // one fs2_op_set_gemm_variant / fs2_set_tuning value; FS2_ERR_ARG for an undefined one
inline const Tuning& tuning_of(const Tuning* t) { return t ? *t : op_tuning(); }

struct GemmArgs {
    const void* X;      // (M, ldx) activations, row-major
    const void* W;      // (N, K) weights, K = taps*Cin (tap-major)
    const float* bias;  // (N) or null
    void* C;            // (M, ldc)
    void* C_lo = nullptr;  // fp32 launches only: C and C_lo receive the result as TWO bf16 (M, ldc) tensors, head and tail (x = hi + lo up
                           // to 2^-17 |x|) - the operands of the split-arithmetic attention (attention.hip, X3); slab kernel, plain epilogue
    int M, N, K;
    int ldx, ldc;
    int Cin, taps, pad;  // implicit conv: K index = tap*Cin + c, source row = m + tap - pad
    int S;               // rows per utterance (zero padding does not cross utterances)
    int relu;
    // Optional fused row epilogue (needs a whole output row in one workgroup: N <= 256 on the slab
    // kernel; otherwise the launcher writes the GEMM result to ln_tmp and runs layernorm_kernel):
    //   y = LayerNorm(act(acc + bias) [+ res]) * ln_g + ln_b ;  C = y (skipped if C == null)
    //   pred[m] = mask[m] ? 0 : sum_n y[m,n] * dot_w[n] + dot_b     (if dot_w)
    const void* res = nullptr;      // (M, ldc) residual, same dtype as X
    const float* ln_g = nullptr;    // (N)  non-null enables the fused epilogue
    const float* ln_b = nullptr;
    float ln_eps = 1e-5f;
    const float* dot_w = nullptr;   // (N)
    float dot_b = 0.f;
    const uint8_t* mask = nullptr;  // (M) 1 = pad
    float* pred = nullptr;          // (M)
    void* ln_tmp = nullptr;         // (M, ldc) scratch for the unfused fallback
    void* z_out = nullptr;          // (M, ldc), C's dtype: the fused epilogue also stores the PRE-norm rows act(acc + bias) [+ res] (the
                                    // training tape: LayerNorm's backward needs them); slab kernel only, null otherwise
    int xcd_remap = 0;              // set by the launcher
    const Tuning* tune = nullptr;   // null: the calling thread's (op_tuning())
    // Deferred-LayerNorm epilogue of the slab kernel (no ln_g): C = v = act(acc + bias) + res, and per row the partial
    // sums (sum v, sum v^2) of every 64-column wave slice -> stats_out (M, ceil(N/256)*4) float2.  epi_res_stats != null:
    // the residual is a pre-norm tensor normalised on load from ITS parts (epi_res_parts per row) with epi_res_g / _b.
    const void* epi_res = nullptr;
    const float* epi_res_stats = nullptr;
    const float* epi_res_g = nullptr;
    const float* epi_res_b = nullptr;
    int epi_res_parts = 0;
    float* stats_out = nullptr;
    int w_presplit = 0;             // split launches: W is NOT fp32 but heads + tails, packed per 32-channel chunk by pack_presplit_weights (gemm_mfma.hip)
    int split = 0;                  // fp32 operands only: 1 = bf16 x 3 split arithmetic in the slab kernel (gemm_mfma.hip)
    const uint8_t* zero_rows = nullptr;  // (M) 1 = store zeros for this row (128x128 kernel only: the mel head)
    const void* gate = nullptr;     // slab kernel, plain epilogue: C = gate > 0 ? gate_scale * (acc + bias) : 0; gate has C's shape, ldc and dtype
    float gate_scale = 1.f;
    float drop_p = 0.f;             // slab kernel, fused LayerNorm epilogue only: z = dropout(act(acc + bias)) [+ res] with the counter-based
    uint64_t drop_seed = 0, drop_key = 0;  // mask of fs2_op_dropout over the (M, N) product (element index row * N + col) - the residual sites of the training step
    // Row-scaled product (plain epilogue of the slab / persistent kernels, no ReLU): X holds PRE-norm rows v whose LayerNorm the
    // caller folded into W and bias (W' = W diag(gamma), bias' = bias + W beta, rs_wg[n] = sum_k W'[n][k] as stored):
    //   C[m][n] = rstd[m] * acc - (rstd[m] * mean[m]) * rs_wg[n] + bias'[n]   =  (LayerNorm(v)[m] . W[n]) + bias[n]
    // rs_stats = (M) float2 (rstd, rstd * mean) per row, finished from the deferred epilogue's parts by launch_rowstats_finish.
    const float* rs_stats = nullptr;
    const float* rs_wg = nullptr;
    // Head behind the deferred LayerNorm (persistent kernel only; with stats_out, C unused): the rows are not stored, per row and
    // 256-column tile sum_n v[m][n] * head_gw[n] -> head_out (M, ceil(N/256)) beside the statistics; launch_head_finish turns both
    // into pred[m] = mask[m] ? 0 : (LayerNorm(v)[m] . w) + b with head_gw = gamma * w.
    const float* head_gw = nullptr;
    float* head_out = nullptr;
    int ksplit = 0;                 // > 1: split-K on the slab kernel (plain epilogue, fp32 out, no bias / ReLU / gate): split s sums the channel
                                    // blocks [s, s + 1) * Cin / ksplit of every tap into plane s of C (ksplit, M, ldc); launch_split_k_reduce adds them
};
int launch_gemm(const GemmArgs& a, int in_dtype, int out_dtype, hipStream_t stream);
// fp32 (N, K) weights, K % 32 == 0 -> the split arithmetic's load-time format, same size: per row and 32-channel chunk (128 bytes)
// 16-byte slot s < 4 = bf16 heads of channels [4s, 4s + 4) and [16 + 4s, 16 + 4s + 4) of the chunk, slot 4 + s = their tails
// (x = hi + lo up to 2^-17 |x|; the values split_bf16x3 produces).  Once per weight tensor, at fs2_finalize.
int launch_presplit_pack(const float* w, void* out, size_t n_values, hipStream_t stream);  // device side; out == w packs in place
bool gemm_presplit_eligible(int N, int K);
// out (M, N) in out_dtype = [out +] sum over s of part[s] (fp32 planes of n = M * N elements, n % 4 == 0), fixed order
int launch_split_k_reduce(const float* part, void* out, size_t n, int ksplit, int accumulate, int out_dtype, hipStream_t stream);
// the split a long-K, few-tile GEMM / conv is worth (1 = none): tools/bench_ops.py dgrad
int gemm_splitk_choice(int M, int N, int Cin, int taps, int S, int in_dtype);
bool gemm_wres_supported(const GemmArgs& a, int in_dtype, int out_dtype, bool force);
int launch_gemm_wres(const GemmArgs& a, hipStream_t stream);
// gemm_persist.hip: the slab kernel's persistent form (one workgroup per CU walks its tiles; bit-identical results)
bool gemm_persist_supported(const GemmArgs& a, int in_dtype, int out_dtype, int mi);
bool gemm_head_supported(const GemmArgs& a, int in_dtype, int out_dtype);  // can this launch take GemmArgs::head_out (persistent kernel, switch on)
bool gemm_persist_pays(const GemmArgs& a, int mi);
int launch_gemm_persist(const GemmArgs& a, int mi, hipStream_t stream);

struct AttnArgs {
    const void* qkv;        // (B*S, 3H): [q | k | v] columns, head h at h*d
    void* vt;               // (B*heads, d, Spad) scratch: V^T, zero padded (written by the launcher)
    const uint64_t* kbits;  // (B, nw64) bit k of word w set <=> key w*64+k is a valid (unpadded) key
    void* out;              // (B*S, H)
    int B, S, H, heads, Spad, nw64;
    float scale_log2e;      // log2(e) / sqrt(d)
    const void* qkv_lo = nullptr;  // split arithmetic (attention_kernel<.., X3>): qkv = the bf16 heads, qkv_lo = the bf16 tails of the fp32
                                   // (B*S, 3H) tensor, out = fp32 rows
    // training path: per-query log-sum-exp in log2 units of the scaled scores (lse2 = m + log2(l); P = exp2(s - lse2)) for the
    // recomputing backward, and the attention-weight dropout of nn.MultiheadAttention (mask over the (b, head, q, key) index)
    float* lse2 = nullptr;  // (B, heads, S) or null
    float drop_p = 0.f;
    uint64_t drop_seed = 0, drop_key = 0;
    const Tuning* tune = nullptr;
};
// r06: encoder-side fused launch (attention.hip attn_out_ln_kernel): self-attention of both heads + out-projection + residual + LayerNorm,
// bf16, H = 256, two heads of 128: out = LayerNorm(res + attention(qkv) W_o^T + bias).  wpk = W_o (256 x 256) in the single-launch predictor's
// fragment order (launch_pack_predictor_weights(w, out, stream, 1)).
struct AttnOutArgs {
    const void* qkv;        // (B*S, 3H) bf16 [q | k | v]
    const uint64_t* kbits;  // (B, nw64) valid-key words
    const void* wpk;        // packed W_o
    const float* bias;      // (H)
    const void* res;        // (B*S, H) bf16 residual
    const float* ln_g;
    const float* ln_b;
    void* out;              // (B*S, H) bf16; may alias res
    int B, S, H, heads, nw64;
    float scale_log2e, eps;
};
bool attn_out_ln_supported(int dtype, int H, int heads, int S);
int launch_attn_out_ln(const AttnOutArgs& a, hipStream_t stream);
// Recomputing (flash) backward of the same attention, bf16, head dim 128 (attention_bwd.hip): dqkv (B*S, 3H) from dout (B*S, H),
// the forward's qkv / lse2 and delta = fs2 attn_delta(dout, out).  Two launches: (dK, dV) per key block, dQ per query block.
struct AttnBwdArgs {
    const void* qkv;
    const void* dout;
    const float* lse2;       // (B, heads, S)
    const float* delta;      // (B, heads, S)
    const uint8_t* key_pad;  // (B, S) 1 = pad
    void* dqkv;              // (B*S, 3H) out: every element written
    int B, S, H, heads;
    float scale_log2e, scale;
    float drop_p;
    uint64_t drop_seed, drop_key;
    const Tuning* tune = nullptr;
};
bool attention_bwd_supported(int dtype, int H, int heads);
int launch_attention_bwd(const AttnBwdArgs& a, int dtype, hipStream_t stream);
int launch_transpose_v(const AttnArgs& a, int dtype, hipStream_t stream);  // fills a.vt from a.qkv
int launch_split_hi_lo(const float* x, void* hi, void* lo, size_t n, hipStream_t stream);  // fp32 -> bf16 head + bf16 tail, n % 8 == 0
int launch_attention(const AttnArgs& a, int dtype, hipStream_t stream);    // needs a.vt filled
// attention_pipe.hip: the software-pipelined kernel for the MFMA-bound instance (bf16, head dim 128, no attention dropout)
bool attention_pipe_supported(const AttnArgs& a, int dtype);
int launch_attention_pipe(const AttnArgs& a, int variant, hipStream_t stream);

// Whole dense VariancePredictor (n x [conv k=3 -> ReLU -> LN] -> Linear(H,1) -> mask) in one launch;
// bf16, H = 256, k = 3 (predictor_fused.hip).  wpk = per-layer weights in MFMA fragment order
// (launch_pack_predictor_weights), bias / ln_g / ln_b = (nlayers, H) fp32.
struct PredictorArgs {
    const void* x;          // (B*S, H) bf16
    const void* wpk;        // nlayers * predictor_packed_bytes_per_layer()
    // depth-wise layers (model.py:541-558; r06): dw_w = (nlayers, 3, H) fp32 tap-major depth-wise weights, dw_b = (nlayers, H); wpk then holds the
    // POINTWISE weights (one tap per layer, predictor_packed_bytes_per_layer(1)) and bias the pointwise bias; taps stays 3 (the depth-wise k)
    const float* dw_w = nullptr;
    const float* dw_b = nullptr;
    const void* wpk_lo = nullptr;  // split-arithmetic form (launch_predictor_fused_x3): wpk = the weights' bf16 heads, wpk_lo their tails; x is fp32
    const float* bias;
    const float* ln_g;
    const float* ln_b;
    const float* head_w;    // (H)
    float head_b;
    const uint8_t* mask;    // (B*S) 1 = pad -> pred 0, or null
    float* pred;            // (B*S)
    int B, S, H, nlayers, taps;
    float eps;
    // Optional tail (inference engine): the VarianceEncoder's  y = x + Embedding[bucketize(pred * std + mean)] [+ pe] [+ spk]
    // (model.py:434-438,333; rowops.hip bucket_embed_kernel, same arithmetic bit for bit) for the rows this workgroup finishes,
    // instead of a launch of its own behind this one.  be_y != x (neighbouring workgroups still read x's rows as their halo).
    void* be_y = nullptr;            // (B*S, H) bf16 out, or null: no tail
    const float* be_bins = nullptr;  // (be_nbins - 1) sorted edges, be_nbins - 1 <= 512
    const float* be_emb = nullptr;   // (be_nbins, H) fp32
    int be_nbins = 0;
    float be_std = 1.f, be_mean = 0.f;
    const float* be_pe = nullptr;    // (>= S, H) fp32 or null
    const float* be_spk = nullptr;   // (B, H) fp32 or null
};
bool predictor_fused_supported(int dtype, int H, int taps, int nlayers, int S);
size_t predictor_packed_bytes_per_layer(int taps = 3);
int launch_pack_predictor_weights(const void* w_layer /*(H, taps*H) tap-major bf16*/, void* out_layer, hipStream_t stream, int taps = 3);
int launch_predictor_fused(const PredictorArgs& a, hipStream_t stream);
bool predictor_fused_x3_supported(int H, int taps, int nlayers, int S);
int launch_predictor_fused_x3(const PredictorArgs& a, hipStream_t stream);

// One Conv1d of the HiFi-GAN generator (vocoder_conv.hip).  Activations are (B, S, channels)
// time-major in the engine dtype; w is the layer's weights in MFMA fragment order
// [n-tile][step = tap * (cin_pad / KE) + kc, padded to a multiple of 4][wn][2][64] x 16 B.
struct VocConvArgs {
    const void* x;          // (B, S, cin) engine dtype, or fp32 when in_fp32 (the mel)
    const void* w;
    const float* bias;      // (n)
    const void* res;        // (B, S, n) or null: added before scaling
    void* out;              // (B, S, n); post: (B, S) fp32
    const int32_t* lengths; // (B) valid FRAMES per utterance or null (= all S rows valid)
    int len_scale;          // rows of this layer per frame
    int B, S;
    int cin, cin_pad, n, taps, dil, pad;  // out[t] = sum_tap x[t - pad + tap*dil] . W[tap]
    int wn;                 // wave columns (n-tile = wn * 32 channels), 8 / wn wave rows
    float in_slope;         // LeakyReLU slope applied to the input while staging (1 = none)
    float scale;            // (acc + bias + res) * scale
    int accumulate;         // += previous contents of out
    int in_fp32;
    int post;               // conv_post: one channel, tanh, fp32 out
    float out_slope = 1.f;  // store LeakyReLU(result): every consumer is a resident resblock launch with x_act
    const Tuning* tune = nullptr;
    int shift_from = 0;     // output channels >= shift_from read their taps one row further on (x[t - pad + 1 + tap*dil]):
                            // the two tap windows of a transposed conv's phases (vocoder_engine.hip up_layer); 0 = none
};
// A whole ResBlock "1" (npairs = 3 (c1 dilated, c2) pairs) or one pair (npairs = 1) on an LDS-resident
// tile, vocoder_resblock.hip.  out = (x after the pairs) * scale (+ previous contents).
struct VocResblockArgs {
    const void* x;          // (B, S, C) block input
    void* out;              // (B, S, C): (resblock(x)) * scale (+ previous contents)
    const void* w;          // 2*npairs convs back to back, fragment order [conv][step][wn][2][64] x 16 B
    const float* bias;      // (2*npairs, C): c1, c2 of each pair in turn
    const int32_t* lengths;
    int len_scale;
    int B, S, C, taps, wn, npairs;
    int dil[3];             // dilation of c1 of each pair (c2 is undilated)
    float slope, scale;
    int accumulate;
    int x_act = 0;          // x already holds lrelu(x) (written by a launch with out_act): the fill is a plain LDS-DMA copy
    int out_act = 0;        // store lrelu(result) for such a consumer (pairs inside a block; never with accumulate)
    const Tuning* tune = nullptr;
};
struct LossArgs {
    const float* pred;   // (rows, inner) fp32
    const void* truth;   // truth_kind 0: fp32 (rows, inner); 1: int64 durations, compared as log(d + 1)
    const uint8_t* mask; // (rows) 1 = pad (excluded)
    void* ws;            // masked_loss_ws_bytes() bytes, zeroed once by the caller
    float* out;          // [mean over the selected elements, number of selected elements]
    int64_t rows;
    int inner, kind, truth_kind;  // kind 0 = l1, 1 = mse
};
struct SoftDtwArgs {
    const float* x;   // (B, N, D) fp32
    const float* y;   // (B, M, D) fp32
    float* out;       // (B) soft-DTW value R[N, M]
    int B, N, M, D;
    float gamma;
};
size_t soft_dtw_lds_bytes(int N, int M, int D);
int launch_soft_dtw(const SoftDtwArgs& a, hipStream_t stream);
size_t soft_dtw_grad_scratch_bytes(int B, int N, int M);
int launch_soft_dtw_grad(const SoftDtwArgs& a, float* grad_x, void* scratch, size_t scratch_bytes, hipStream_t stream);
size_t masked_loss_ws_bytes();
int launch_masked_loss(const LossArgs& a, hipStream_t stream);
int voc_resblock_mi16(const VocResblockArgs& a, int dtype);  // 0 = shape not covered
int launch_vocoder_resblock(const VocResblockArgs& a, int dtype, hipStream_t stream);
int voc_steps_padded(int taps, int cin_pad, int dtype);
int launch_vocoder_conv(const VocConvArgs& a, int dtype, hipStream_t stream);

struct ConvertArgs {
    const void* src;
    void* dst;
    size_t n;
};
int launch_convert(const ConvertArgs& a, int src_dtype, int dst_dtype, hipStream_t stream);

struct LayerNormArgs {
    const void* x;        // (M, H)
    const void* res;      // (M, H) or null: y = LN(x + res)
    const float* gamma;   // (H)
    const float* beta;    // (H)
    void* y;              // (M, H) or null (skip the store when only the row-dot is needed)
    const float* dot_w;   // (H) or null: pred[m] = mask[m] ? 0 : sum_c y[m,c]*dot_w[c] + dot_b
    float dot_b;
    const uint8_t* mask;  // (M) 1 = pad, or null
    float* pred;          // (M)
    int M, H;
    float eps;
    const float* pre_stats = nullptr;  // (M, pre_parts) float2 partial (sum, sum of squares) of x's rows: skip the reductions
    int pre_parts = 0;
    const float* dot_b_dev = nullptr;  // device scalar used instead of dot_b when given (a trained bias: no host read-back)
    float drop_p = 0.f;                // > 0: y = dropout(LN(.)), mask of fs2_op_dropout over the (M, H) element index
    uint64_t drop_seed = 0, drop_key = 0;
};
int launch_layernorm(const LayerNormArgs& a, int dtype, hipStream_t stream);

// (M, nparts) float2 partial (sum, sum of squares) over ncols columns per row -> (M) float2 (rstd, rstd * mean): the per-row
// constants of the row-scaled GEMM epilogue (GemmArgs::rs_stats), once per tensor instead of once per column tile of its consumer
int launch_rowstats_finish(const float* parts, int nparts, int ncols, float eps, float* out, int M, hipStream_t stream);
// pred[m] = mask[m] ? 0 : rstd (sum of dots[m][:] - mean * sum_gw) + cst from the deferred epilogue's parts (GemmArgs::head_out)
int launch_head_finish(const float* parts, const float* dots, int nparts, int ncols, float eps, float sum_gw, float cst, const uint8_t* mask, float* pred,
                       int M, hipStream_t stream);

struct DwConvArgs {
    const void* x;      // (B*S, C)
    const float* w;     // (C, k)
    const float* bias;  // (C) or null
    void* y;            // (B*S, C)
    int B, S, C, k, pad;
    // x is a pre-norm tensor: LayerNorm it on load from its row parts (ln_parts float2 per row), gamma / beta (C)
    const float* ln_stats = nullptr;
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    int ln_parts = 0;
    float ln_eps = 1e-5f;
    int flip = 0;        // 1: taps in reverse order (the data gradient of the same conv: dx = dwconv(dy, flipped w), no bias)
};
int launch_dwconv(const DwConvArgs& a, int dtype, hipStream_t stream);

// CWT pitch head (VarianceEncoder.forward CWT branch, model.py:412-431 + CWT.recompose, dataset/cwt.py:18-21,49-50)
struct CwtArgs {
    const void* out_conv;   // (B*T, F) last predictor layer's LayerNorm output, engine dtype
    const float* spec;      // (B*T, ld_spec) head output: the first 10 columns are the wavelet scales
    int ld_spec;
    const uint8_t* mask;    // (B*T) 1 = pad: spectrogram row := 0 (model.py:515-518)
    const float* ms_w;      // (2, F) mean_std_linear
    const float* ms_b;      // (2)
    float* mean_std;        // (B, 2) out
    float* pred;            // (B, T) out: z-normalised sum over the scales * std + mean (log domain)
    float* spec_out;        // (B, T, 10) out or null
    int B, T, F;
};
int launch_cwt_head(const CwtArgs& a, int dtype, hipStream_t stream);

struct EmbedArgs {
    const int64_t* phones;  // (B, L)
    const float* table;     // (n_phones, H), row 0 == 0
    const float* pe;        // (>=L, H)
    const float* spk;       // (B, H)
    void* x;                // (B*L, H)
    uint8_t* src_mask;      // (B, L) 1 = pad
    int B, L, H, n_phones;
};
int launch_embed(const EmbedArgs& a, int dtype, hipStream_t stream);

struct SpkProjArgs {
    const float* dvec;  // (B, Din)
    const float* w;     // (H, Din)
    const float* b;     // (H)
    float* spk;         // (B, H) = relu(W dvec + b)
    int B, H, Din;
};
int launch_spk_proj(const SpkProjArgs& a, hipStream_t stream);

struct MaskBitsArgs {
    const uint8_t* mask;  // (B, S) 1 = pad
    uint64_t* bits;       // (B, nw64)
    int B, S, nw64;
};
int launch_mask_bits(const MaskBitsArgs& a, hipStream_t stream);

struct DurationArgs {
    const float* dur_pred;    // (B, L) log-domain prediction, 0 at pads
    const uint8_t* src_mask;  // (B, L)
    const int32_t* forced;    // (B, L) or null: use these durations (no rounding, no guard)
    int32_t* dur;             // (B, L) out
    int32_t* cum;             // (B, L) inclusive prefix sum
    int32_t* totals;          // (B)
    int32_t* guard;           // (B) 1 if the zero-duration guard fired
    int B, L;
};
int launch_durations(const DurationArgs& a, hipStream_t stream);

struct RegulateArgs {
    const void* x;            // (B*L, H)
    const int32_t* cum;       // (B, L)
    const int32_t* totals;    // (B)
    void* y;                  // (B*T, H)
    uint8_t* tgt_mask;        // (B, T) 1 = pad (t >= total, untruncated)
    int B, L, T, H;
};
int launch_regulate(const RegulateArgs& a, int dtype, hipStream_t stream);

struct BucketArgs {
    const void* x;        // (B*T, H)
    const float* pred;    // (B*T) or null (no embedding: only + pe + spk)
    const float* bins;    // (nbins-1)
    const float* emb;     // (nbins, H)
    int nbins;
    float std, mean;
    const float* pe;      // (>=T, H) or null
    const float* spk;     // (B, H) or null
    void* y;              // (B*T, H) (may alias x)
    int32_t* idx_out;     // (B*T) or null: bucket indices (debug / parity)
    int B, T, H;
    const int32_t* forced_idx;  // (B*T) or null: use these bucket indices instead of searching
    int pred_per_utt = 0;       // 1: pred has one value per utterance (B), broadcast over T
    const float* bucket_src = nullptr;  // (B*T) or null: bucketize THIS (teacher-forced target) instead of pred
};
int launch_bucket_embed(const BucketArgs& a, int dtype, hipStream_t stream);

// ---- training step: backward kernels ------------------------------------------------------------
// General strided-batched GEMM (bgemm.hip): C[b1][b2](m, n) = alpha * sum_k A(m, k) B(k, n) + bias[n] + beta * C.
struct BGemmArgs {
    const void* A;
    const void* B;
    void* C;
    const float* bias = nullptr;  // (N) or null
    int M = 0, N = 0, K = 0;
    long sAm = 0, sAk = 0;        // element strides of A(m, k); one of them is 1
    long sBk = 0, sBn = 0;        // element strides of B(k, n); one of them is 1
    long ldc = 0;                 // C is n-contiguous
    int nb1 = 1, nb2 = 1;         // two batch levels (e.g. utterance, head)
    long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sC1 = 0, sC2 = 0;
    float alpha = 1.f, beta = 0.f;
    int splitk = 1;               // > 1: per-split slabs in ws (bgemm_ws_bytes) + a reduce pass; deterministic
    float* ws = nullptr;
    // implicit 'same'-padded Conv1d over (B*S, C) time-major rows, utterances of `seg` rows (0 = plain GEMM):
    int seg = 0;
    int taps = 1, Kin = 0;        // dgrad form: K = taps * Kin; k-tile of tap j reads A rows m + a_shift0 + j * a_shift_step
    int a_shift0 = 0, a_shift_step = 0;
    long sBtap = 0;               //             ... and B from + j * sBtap
    int b_shift0 = 0, b_shift_step = 0;  // wgrad form (taps == 1): batch index b2 reads B's k rows at k + b_shift0 + b2 * b_shift_step
    // fused softmax backward (attention): C = alpha * P * (dropout(acc) - delta[b1][b2][m]); P has C's layout and dtype
    const void* epi_p = nullptr;
    const float* epi_delta = nullptr;   // (nb1, nb2, M)
    float drop_p = 0.f;                 // attention-weight dropout of the forward (mask over the (nb1, nb2, M, N) index)
    uint64_t drop_seed = 0, drop_key = 0;
    int c_dtype = FS2_F32;        // dtype of C: bf16 operands may write bf16 (activations) or fp32 (weight gradients, scores)
    int vecA = 0, vecB = 0;       // set by the launcher
    int xcd_remap = 0;            // set by the launcher
    const Tuning* tune = nullptr;
};
size_t bgemm_ws_bytes(const BGemmArgs& a);
bool bgemm_tn256_eligible(const BGemmArgs& a);  // bf16 operands only
int launch_bgemm(const BGemmArgs& a, int dtype, hipStream_t stream);

// Row / column kernels of the backward (backward.hip).  T = activation dtype; gradients of parameters are fp32.
struct LayerNormBwdArgs {
    const void* z;        // (M, H) the tensor that was normalised ...
    const void* res;      // (M, H) or null: ... plus this (the forward's y = LN(x + res))
    const void* dy;       // (M, H)
    const float* gamma;   // (H)
    void* dz;             // (M, H) out
    float* part;          // (nparts, 3, H) out: per-workgroup partial column sums of dy * zhat, dy and dz (-> col_sum -> dgamma,
                          // dbeta, and the bias gradient of the linear / conv layer whose output z is)
    int M, H, nparts;
    float eps;
    int relu_mask;        // 1: z is a ReLU output; dz := dz where z > 0 else 0 (the gradient of the PRE-activation)
    float drop_p = 0.f;   // > 0: dy is the gradient of dropout(y): the mask of the forward (seed, key) is applied to dy on load
    uint64_t drop_seed = 0, drop_key = 0;
    // second output: dz through the dropout that sat on ONE summand of z (z = res + dropout(u): dzm = mask o dz / (1 - p) = du, while dz
    // itself is the residual's gradient).  With it the third column sum of `part` is that of dzm (u's bias gradient).
    void* dzm = nullptr;  // (M, H) out or null
    float out_p = 0.f;
    uint64_t out_key = 0; // (seed = drop_seed is shared: one seed per micro-step)
};
int layernorm_bwd_parts(int M);
int launch_layernorm_bwd(const LayerNormBwdArgs& a, int dtype, hipStream_t stream);

struct ColSumArgs {
    const void* x;    // (M, N) in the launch dtype, row stride ldx
    float* out;       // (nseg, N): out[s][n] (+)= scale * sum over the rows of segment s
    float* ws;        // col_sum_ws_bytes; its first 8192 words are counters: zero before the first launch, left zero by every launch
    int M, N, ldx;
    int seg;          // rows per segment (0 or M: one segment)
    int accumulate;   // 1: out += ...
    float scale;
    float* out2 = nullptr;  // columns n >= n1 go to out2[s][n - n1] instead (out is then n1 wide)
    int n1 = 0;
    int accumulate2 = 0;
    const float* row_w = nullptr;  // (M) or null: rows are weighted, out[s][n] (+)= scale * sum_r row_w[r] x[r][n]
    const Tuning* tune = nullptr;
};
size_t col_sum_ws_bytes(int M, int N, int seg);
int launch_col_sum(const ColSumArgs& a, int dtype, hipStream_t stream);

struct SoftmaxArgs {
    const float* s;         // (B, heads, S, S) fp32: scores (fwd) / dP (bwd)
    const void* p;          // bwd: the probabilities, activation dtype
    void* out;              // fwd: probabilities; bwd: dS; activation dtype (may alias s when that is fp32)
    const uint8_t* key_pad; // (B, S) 1 = pad key -> probability 0; fwd only
    int B, heads, S;
    float scale;            // fwd: softmax(scale * s); bwd: dS = scale * P o (dP - sum(dP o P))
};
int launch_softmax_fwd(const SoftmaxArgs& a, int dtype, hipStream_t stream);
int launch_softmax_bwd(const SoftmaxArgs& a, int dtype, hipStream_t stream);

struct EwArgs {   // elementwise helpers over n elements of the launch dtype
    const void* a;
    const void* b;
    void* out;
    size_t n;
    float alpha, beta;
    int op;       // 0: out = alpha*a + beta*b   1: out = a * (b > 0)  (ReLU backward: b = the ReLU output)   2: out = alpha * a
};
int launch_ew(const EwArgs& a, int dtype, hipStream_t stream);

struct ScatterRowsArgs {     // table[idx[r]] += scale * x[r]  (embedding backward), deterministic: one workgroup per table row
    const void* x;           // (R, H) launch dtype
    const int32_t* idx32;    // (R) or null
    const int64_t* idx64;    // (R) or null
    float* table;            // (V, H) accumulated into
    int R, H, V;
    int skip_row;            // table row that receives no gradient (padding_idx), -1 = none
    float* ws = nullptr;     // scatter_rows_ws_bytes (0: not needed) or null: the one-launch kernel
};
size_t scatter_rows_ws_bytes(int R, int H, int V);
int launch_scatter_rows(const ScatterRowsArgs& a, int dtype, hipStream_t stream);

struct RegulateBwdArgs {     // d_phone[b][p] = sum of d_frame[b][t] over the frames phone p was repeated to (t < T)
    const void* dy;          // (B*T, H) launch dtype
    const int32_t* cum;      // (B, L) inclusive prefix sums of the durations
    void* dx;                // (B*L, H)
    int B, L, T, H;
};
int launch_regulate_bwd(const RegulateBwdArgs& a, int dtype, hipStream_t stream);

struct LossBwdArgs {         // gradient of alpha * mean over selected elements of |p - t| or (p - t)^2
    const float* pred;       // (rows, inner)
    const void* truth;       // as LossArgs
    const uint8_t* mask;     // (rows) 1 = pad
    const float* stat;       // the forward's [mean, count] (device): count is read here, no host round trip
    float* dpred;            // (rows, inner)
    int64_t rows;
    int inner, kind, truth_kind;
    float alpha;
};
int launch_masked_loss_bwd(const LossBwdArgs& a, hipStream_t stream);

// depth-wise Conv1d weight / bias gradient: part[chunk][c * k + j] = sum_t dy[t][c] x[t + j - pad][c], part[chunk][C * k + c] =
// sum_t dy[t][c] over the chunk's rows (dwconv_wgrad_parts(B, S) chunks; -> col_sum -> dw (C, k), db (C))
struct DwConvWgradArgs {
    const void* dy;   // (B*S, C)
    const void* x;    // (B*S, C)
    float* part;      // (nparts, C * (k + 1))
    int B, S, C, k, pad;
};
int dwconv_wgrad_parts(int B, int S);
int launch_dwconv_wgrad(const DwConvWgradArgs& a, int dtype, hipStream_t stream);

// conv2 = Sequential(grouped 1x1 conv over F channels in H groups, pointwise F -> H) of the depth-wise ConformerEncoderLayer
// (model.py:84-93) as ONE linear map: Wf[o][g*gs + j] = sum_i W21[o][g*gs + i] G[g*gs + i][j], bf = b21 + W21 bg; and the
// chain rule back from (dWf, dbf) to the four parameter gradients (accumulating).
struct FoldConv2Args {
    const float* G;    // (F, gs)
    const float* bg;   // (F)
    const float* W21;  // (H, F)
    const float* b21;  // (H)
    void* Wf;          // (H, F) in wf_dtype
    float* bf;         // (H)
    int H, F;
};
int launch_fold_conv2(const FoldConv2Args& a, int wf_dtype, hipStream_t stream);
struct UnfoldConv2Args {
    const float* dWf;  // (H, F) fp32
    const float* dbf;  // (H)
    const float* G; const float* bg; const float* W21;
    float* dG; float* dbg; float* dW21; float* db21;   // accumulated into
    int H, F;
};
int launch_unfold_conv2(const UnfoldConv2Args& a, hipStream_t stream);

struct DropoutArgs {   // y = x * keep / (1 - p), keep from a counter-based hash of (seed, key, element index); y may alias x
    const void* x;
    void* y;
    size_t n;
    float p;
    uint64_t seed, key;
};
int launch_dropout(const DropoutArgs& a, int dtype, hipStream_t stream);
struct AttnDeltaArgs {   // delta (B, heads, S) = per-head row dot of dout and out, both (B*S, H)
    const void* dout;
    const void* out;
    float* delta;
    int B, S, H, heads;
};
int launch_attn_delta(const AttnDeltaArgs& a, int dtype, hipStream_t stream);
struct RowDotArgs {    // pred[m] = mask[m] ? 0 : y[m] . w + b[0]
    const void* y;
    const float* w;
    const float* b;
    const uint8_t* mask;
    float* pred;
    long M;
    int H;
};
int launch_row_dot(const RowDotArgs& a, int dtype, hipStream_t stream);

struct TransposeWeightArgs {   // dst (Cin, taps*N) [ci][j'*N + n] = src (N, taps*Cin) [n][(taps-1-j')*Cin + ci]
    const void* src;
    void* dst;
    int N, Cin, taps;
};
int launch_transpose_weight(const TransposeWeightArgs& a, int dtype, hipStream_t stream);
int launch_transpose_weight_batch(const long long* tab, int n, long long tiles, hipStream_t stream);  // bf16; tab on the device

struct AdamWArgs {
    float* p; const float* g; float* m; float* v;   // flat fp32 buffers of n elements
    size_t n;
    float lr, beta1, beta2, eps, weight_decay;
    int step;                // 1-based
    const float* gnorm_sq;   // device scalar: sum of squares of g (null = no clipping)
    float max_norm;          // clip the global norm to this (gradient_clip_val)
    float grad_scale;        // g is multiplied by this first (1 / accumulated micro-batches)
    void* shadow = nullptr;  // (n) bf16 or null: the updated weights again, rounded to bf16
};
int launch_adamw(const AdamWArgs& a, hipStream_t stream);
size_t sum_sq_ws_bytes(size_t n);
int launch_sum_sq(const float* x, size_t n, float* ws, float* out, hipStream_t stream);

}  // namespace fs2
