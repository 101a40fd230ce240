/* C ABI of the MI355X-native FastSpeech2 / LightSpeech mel forward (libfs2_hip.so).
 *
 * The reference (MiniXC/LightningFastSpeech2) has no FFI layer: its "operator API" for this path is
 * the Python method  FastSpeech2.forward(targets: dict, inference) -> dict
 * (litfass/fastspeech2/fastspeech2.py:636-784) over a state_dict whose key names are fixed by the
 * module tree built at fastspeech2.py:242-438.  This header is what a binding for
 * that method binds (ctypes stub in INTEGRATION.md; the in-tree host mirror is
 * lightningfastspeech2_amd/model.py):
 *
 *   fs2_create / fs2_load_weight / fs2_finalize   <- FastSpeech2.__init__ + load_state_dict
 *                                                    (fastspeech2.py:46-491, :530-620): weights are
 *                                                    passed under the reference's own key names
 *   fs2_encode + fs2_decode                       <- FastSpeech2.forward(batch, inference=True)
 *                                                    (fastspeech2.py:636-731), split at the one point
 *                                                    where the output length T becomes known
 *                                                    (LengthRegulator, model.py:354-355)
 *
 * Plain pointers and sizes only; device pointers are HIP device addresses on the engine's device
 * (e.g. torch tensors' data_ptr()).  All work is enqueued on the caller's stream; no hidden streams,
 * no global state, one engine per device.  Every function returns a status code (0 = OK);
 * fs2_last_error() has the text.  No exceptions cross this boundary.
 */
#ifndef FS2_H_
#define FS2_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS2_ABI_VERSION 4
#define FS2_MAX_LAYERS 32
#define FS2_MAX_VARIANCES 4
#define FS2_MAX_PRIORS 8   /* hparams.priors: the shipped recipe lists five (scripts/train.sh:49: energy duration snr pitch srmr) */
#define FS2_NAME_LEN 32

enum fs2_status {
    FS2_OK = 0,
    FS2_ERR_HIP = 1,     /* a HIP runtime call or kernel launch failed */
    FS2_ERR_SHAPE = 2,   /* shape/config outside what the kernels support */
    FS2_ERR_ARG = 3,     /* null / inconsistent argument */
    FS2_ERR_WEIGHT = 4,  /* unknown weight name, wrong shape, or weights missing at finalize */
    FS2_ERR_STATE = 5,   /* call order violated (e.g. decode before encode) */
    FS2_ERR_NOMEM = 6
};

enum fs2_dtype {
    FS2_F32 = 0,
    FS2_BF16 = 1,
    FS2_MIXED = 2,    /* engine modes only (fs2_config.dtype): fp32 "front" + bf16 "back", see fs2_config.dtype */
    FS2_MIXED_X3 = 3, /* the same with the front's GEMMs / convs as bf16 x 3 split products of fp32 operands */
    FS2_F32_X3 = 4    /* engine mode: F32 storage and row arithmetic everywhere, EVERY GEMM / conv as bf16 x 3 split products */
};

/* Mirrors the hparams that shape FastSpeech2.forward (fastspeech2.py:46-130). */
typedef struct fs2_config {
    int32_t abi_version;   /* FS2_ABI_VERSION */
    int32_t dtype;         /* fs2_dtype: arithmetic mode. F32 = parity mode (fp32 MFMA, <=1e-3 vs the
                              reference); BF16 = throughput mode (bf16 storage + MFMA, fp32 accumulate,
                              fp32 softmax/LayerNorm statistics/predictor heads/bucketize);
                              MIXED = decision-safe throughput mode: everything a discrete decision depends on
                              (embedding, encoder, duration predictor and rounding, length regulator, variance
                              predictors and bucketize - model.py:259,299-309,315-333,434-438) runs as in F32, the
                              decoder and the mel head (where an error stays an error of O(bf16 rounding) in the
                              mel) as in BF16: durations / buckets follow the fp32 path's, at bf16 decoder speed;
                              MIXED_X3 = MIXED with the front's matrix products evaluated as hi*hi + hi*lo + lo*hi of
                              bf16 head/tail pairs split from the fp32 operands in registers (fp32 storage,
                              accumulation, attention, LayerNorm, heads; ~1e-5 relative per product instead of fp32's
                              6e-8 or bf16's 4e-3): 3 bf16 MFMAs per 32 k-values instead of 8 fp32 MFMAs;
                              F32_X3 = F32 with every GEMM / conv (both sides) evaluated that way: the parity mode's layout,
                              attention, LayerNorm, heads and decisions logic at about half its time (C2: 8.1 vs 15.2 ms per
                              forward); measured 1.4e-5 on the mel against F32 under equal decisions */
    int32_t n_phones;      /* len(phone2id) */
    int32_t hidden;        /* encoder_hidden == decoder_hidden */
    int32_t n_mels;
    int32_t dvec_dim;      /* 256 */
    int32_t max_frames;    /* int(max_length * sampling_rate / hop_length) = 2756 */
    int32_t pe_len;        /* rows of positional_encoding.pe (5000) */
    int32_t enc_layers, enc_heads, enc_filter, enc_depthwise;
    int32_t enc_kernels[FS2_MAX_LAYERS];
    int32_t dec_layers, dec_heads, dec_filter, dec_depthwise;
    int32_t dec_kernels[FS2_MAX_LAYERS];
    int32_t n_variances;   /* hparams.variances in order; level per variance in var_level (ABI v4), transform in var_cwt */
    char var_names[FS2_MAX_VARIANCES][FS2_NAME_LEN];
    int32_t var_nlayers[FS2_MAX_VARIANCES];
    int32_t var_kernel[FS2_MAX_VARIANCES];
    float var_mean[FS2_MAX_VARIANCES];  /* stats[var]["mean"/"std"] (model.py:406-407,434) */
    float var_std[FS2_MAX_VARIANCES];
    int32_t var_filter, var_nbins, var_depthwise;
    int32_t dur_nlayers, dur_kernel, dur_filter, dur_depthwise;
    int32_t n_priors;      /* hparams.priors (default []): PriorEmbedding rows added after the encoder */
    char prior_names[FS2_MAX_PRIORS][FS2_NAME_LEN];
    int32_t var_cwt[FS2_MAX_VARIANCES];  /* variance_transforms[i] == "cwt" (the class default for pitch, fastspeech2.py:60):
                              the CWT head of VarianceEncoder (model.py:412-431,445-461): predictor.linear is (10, filter),
                              mean_std_linear (2, filter) exists, bins are log-spaced; var_mean/var_std must be 0 / 1 */
    int32_t var_level[FS2_MAX_VARIANCES];  /* ABI v4: 0 = frame level (variance_levels[i] == "frame": predicted on the regulated frames,
                              model.py:315-333), 1 = phone level ("phone", model.py:276-294: predicted on the encoder output after the
                              duration predictor has run, its embedding added BEFORE the length regulator; (B, L) outputs) */
} fs2_config;

typedef struct fs2_engine fs2_engine;

/* Caller-owned DEVICE buffers fs2_decode fills (shapes use T returned by fs2_encode).  Any pointer
 * may be NULL to skip that output.  Mask bytes are 1 = pad, directly usable as torch.bool storage. */
typedef struct fs2_outputs {
    float* mel;                    /* (B, T, n_mels) fp32, pad rows included (fastspeech2.py:723) */
    float* duration_prediction;    /* (B, L) log(1+d) domain, 0 at pads (model.py:259,517-518) */
    int32_t* duration_rounded;     /* (B, L) (model.py:299-309) */
    uint8_t* src_mask;             /* (B, L) phones == 0 (fastspeech2.py:651) */
    uint8_t* tgt_mask;             /* (B, T) t >= total_b (model.py:358-361) */
    float* variances[FS2_MAX_VARIANCES];  /* (B, T) each - (B, L) for a phone-level variance -, 0 at pads (model.py:328); for a CWT
                                             variance the recomposed log-domain signal (exp of it = the reference's "reconstructed_signal") */
    float* var_spectrogram[FS2_MAX_VARIANCES];  /* CWT variances only: (B, T, 10) ((B, L, 10) at phone level) wavelet spectrogram, 0 at pads */
    float* var_mean_std[FS2_MAX_VARIANCES];     /* CWT variances only: (B, 2) utterance-level mean, std (model.py:414-415) */
} fs2_outputs;

int fs2_abi_version(void);
const char* fs2_status_string(int status);
const char* fs2_last_error(const fs2_engine* e);

int fs2_create(const fs2_config* cfg, fs2_engine** out);
int fs2_destroy(fs2_engine* e);
/* A second engine over the same device weights (finalized engines only; the weights are freed with the last holder): own workspace,
 * own host-side state.  One engine serves one caller thread at a time; an engine and its clones may run concurrently on different
 * streams - several forwards of FastSpeech2.forward in flight (generate.py:186-195 feeds the model chunk after chunk). */
int fs2_clone(const fs2_engine* src, fs2_engine** out);

/* One call per state_dict entry, reference key names, fp32 HOST data, torch shape.
 * Unknown names that belong to off-path modules are rejected with FS2_ERR_WEIGHT. */
int fs2_load_weight(fs2_engine* e, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* Packs (conv weights tap-major, grouped 1x1 folded into the following pointwise conv), converts to
 * the arithmetic dtype and uploads.  FS2_ERR_WEIGHT if any tensor of the architecture is missing. */
int fs2_finalize(fs2_engine* e);

/* Phase 1: embedding + encoder + duration predictor + rounding/guard + prefix sums.
 *   phones   (B, L) int64 device, 0 = [PAD];  speaker (B, dvec_dim) fp32 device.
 *   forced_durations: NULL, or (B, L) int32 device durations used instead of the predicted ones.
 * Synchronises the stream once to return T = min(max_b sum_l d[b,l], max_frames). */
int fs2_encode(fs2_engine* e, const int64_t* phones, const float* speaker, int32_t B, int32_t L,
               const int32_t* forced_durations, void* hip_stream, int32_t* T_out);
/* Utterance-level priors for the NEXT fs2_encode (one-shot): (n_priors, B) fp32 device values, row p
 * = targets["priors_<prior_names[p]>"] (fastspeech2.py:687-692; PriorEmbedding, model.py:146-164).
 * Required before every fs2_encode when n_priors > 0. */
int fs2_set_priors(fs2_engine* e, const float* priors_device, int32_t B);
/* Per-utterance frame totals (untruncated) and zero-duration-guard flags of the last fs2_encode. */
int fs2_last_totals(const fs2_engine* e, int32_t* totals_host, int32_t* guard_host, int32_t B);
/* Phase 2: length regulator + variance encoders + decoder + mel linear, into caller buffers. */
int fs2_decode(fs2_engine* e, const fs2_outputs* out, void* hip_stream);

/* Workspace.  By default the engine keeps two grow-only device arenas of its own.  A caller that
 * wants every byte to come from its allocator (torch's caching allocator in the host mirror) asks for the sizes
 * and hands the buffers over; the engine then never calls hipMalloc/hipFree on the forward path.
 *   persist: live from fs2_encode to the end of fs2_decode (encoder output, durations, prefix sums); depends on (B, L)
 *   scratch: per-phase; for T = 0 the size fs2_encode needs, for T > 0 max(encode, decode at T frames)
 * fs2_set_workspace(NULL, 0, NULL, 0) returns to engine-owned arenas.  Between fs2_encode and fs2_decode only the
 * scratch buffer may be replaced (the usual pattern: T is known only after fs2_encode).  Buffers 256-byte aligned;
 * a buffer that is too small makes the next phase fail with FS2_ERR_NOMEM and the needed size in fs2_last_error. */
int fs2_workspace_bytes(const fs2_engine* e, int32_t B, int32_t L, int32_t T, size_t* persist_bytes, size_t* scratch_bytes);
int fs2_set_workspace(fs2_engine* e, void* persist_device, size_t persist_bytes, void* scratch_device, size_t scratch_bytes);
/* Data-parallel "global pad" mode: after fs2_encode, raise this shard's frame count to the maximum over
 * all ranks so that fs2_decode pads exactly as the whole batch would (T can only grow, <= max_frames). */
int fs2_set_frames(fs2_engine* e, int32_t T);
/* on = 1: fs2_decode writes zeros into the mel rows of pad frames (t >= the utterance's total) instead of the
 * reference's deterministic pad-row values - what the multi-GPU mel gather ships, fused into the mel GEMM's store. */
int fs2_set_zero_pad_mel(fs2_engine* e, int32_t on);

/* Parity taps: copy an intermediate of the last forward as fp32 into a caller DEVICE buffer.
 * what = "encoder_out" (B,L,H) | "regulated" | "adaptor_out" | "decoder_out" (B,T,H) |
 *        "bucket_<var>" (B,T) int32.  Requires fs2_set_debug(e, 1) before fs2_encode. */
int fs2_set_debug(fs2_engine* e, int32_t on);
/* A/B and parity aid: on = 0 runs every VariancePredictor as per-layer conv+LayerNorm launches, 1
 * (default) as the single-launch kernel where the shape allows it (bf16, filter 256, k = 3, dense). */
int fs2_set_fused_predictor(fs2_engine* e, int32_t on);
/* 1: fs2_decode replays its launches (~50: variance adaptor after the length regulator, decoder, mel head) as a hipGraph once
 * the same shape AND buffer addresses (outputs, arenas) are seen again - first sight runs plainly, second captures, later calls
 * are one hipGraphLaunch on an engine-owned stream ordered against the caller's with events.  Same kernels, same arguments:
 * results are bit-identical.  Off by default; debug taps, per-class profiling and forced buckets always take the
 * plain path; a caller workspace (fs2_set_workspace) is part of the signature.  fs2_graph_replays: how many decodes were replays (tests, diagnostics). */
int fs2_set_graphs(fs2_engine* e, int32_t on);
int64_t fs2_graph_replays(const fs2_engine* e);
/* A/B and parity aid: hidden sizes above 256 with depth-wise blocks (LightSpeech, model.py:73-93,541-558) run LayerNorm
 * deferred (default 1): the GEMM in front of a LayerNorm leaves pre-norm rows + row statistics, the depth-wise conv / the
 * next residual add normalise on load and the remaining LayerNorms are normalise-only passes; 0 = GEMM launch + LayerNorm
 * launch everywhere (the round-1 path). */
int fs2_set_deferred_layernorm(fs2_engine* e, int32_t on);
/* A/B: 1 (default) = inside a stack of wide depth-wise blocks (bf16) a block's closing LayerNorm is not materialised either: the
 * next block's in-projection runs on the pre-norm rows with gamma / beta folded into its weights (fs2_op_gemm_rowscale) and its
 * out-projection normalises the residual on load; 0 = one normalise-only pass per block.  Needs deferred LayerNorm on. */
int fs2_set_folded_layernorm(fs2_engine* e, int32_t on);
/* one kernel-selection switch of THIS engine: the values of fs2_op_set_gemm_variant below; clones made afterwards inherit it */
int fs2_set_tuning(fs2_engine* e, int32_t knob);
/* Parity aid (the analogue of the reference's teacher forcing of variance targets,
 * model.py:417-422): the NEXT fs2_decode embeds these (B, T) int32 device bucket indices for
 * variance `variance_index` instead of bucketizing its own prediction.  One-shot.  For a PHONE-level variance the indices
 * are (B, L) and are consumed by the NEXT fs2_encode (call before it). */
int fs2_force_buckets(fs2_engine* e, int32_t variance_index, const int32_t* idx_device);
/* Teacher forcing as the reference's non-inference forward does it (model.py:317-325,417-422): the
 * NEXT fs2_decode embeds bucketize(target*std + mean) of these (B, T) fp32 device target values for
 * variance `variance_index`; the prediction is still computed and returned.  One-shot.  Together
 * with fs2_encode's forced_durations (= targets["duration"], model.py:296-297) this is
 * FastSpeech2.forward(batch, inference=False) without the loss.  A PHONE-level variance (model.py:278-286) takes (B, L)
 * targets, consumed by the NEXT fs2_encode (call before it). */
int fs2_force_variance_targets(fs2_engine* e, int32_t variance_index, const float* target_device);
int fs2_debug_copy(fs2_engine* e, const char* what, void* dst_device, void* hip_stream);

/* Kernel timing with HIP events on the launch stream (bench.py roofline).  kernel_class: */
enum fs2_kernel_class {
    FS2_K_CONV_GEMM = 0,   /* every implicit-GEMM Conv1d launch (taps > 1) */
    FS2_K_GEMM = 1,        /* pointwise / linear GEMM launches */
    FS2_K_ATTENTION = 2,
    FS2_K_ROWOPS = 3,
    FS2_K_DEC_FFN_CONV1 = 4, /* the decoder FFN's first conv only: the single dominant launch shape */
    FS2_K_DEC_ATTENTION = 5, /* the decoder stack's self-attention launches only: the MFMA-bound attention instance (north_star) */
    FS2_K_ENC_MHA = 6,     /* the encoder stack's whole self-attention block: in-projection + attention + out-projection (+ residual +
                            * LayerNorm) launches - nn.MultiheadAttention inside ConformerEncoderLayer.forward, model.py:108-116 */
    FS2_K_PREDICTOR = 7,   /* the decode phase's single-launch VariancePredictor launches (model.py:482-522 at the frame level): the dominant
                            * kernel of the decision-safe modes, whose matrix work is three bf16 MFMAs per product (r06) */
    FS2_K_COUNT = 8
};
int fs2_profile_enable(fs2_engine* e, int32_t kernel_class, int32_t enable);
/* pre-create the event pairs of `pairs` bracketed launches (otherwise they are created on first use, inside the caller's timed region) */
int fs2_profile_reserve(fs2_engine* e, int32_t kernel_class, int32_t pairs);
/* Sums elapsed ms / launches / algorithmic flops / algorithmic bytes since enable; syncs the events. */
int fs2_profile_read(fs2_engine* e, int32_t kernel_class, double* total_ms, int64_t* launches,
                     double* flops, double* bytes);

/* ---- single-operator entry points (device pointers; used by the parity tests) ------------- */
/* Kernel-selection switches for the parity tests and A/B runs.  They are NOT process state: fs2_op_set_gemm_variant sets a switch of
 * the CALLING THREAD, read by the operator-level entry points below (fs2_op_*) that thread calls; an engine owns its own set
 * (defaults at fs2_create; fs2_set_tuning changes one; fs2_clone copies them) and never looks at a thread's.  An undefined value is
 * FS2_ERR_ARG.  Every non-default form is pinned bit-identical (or within a stated tolerance) to the default by a test.
 *   0       auto (by problem size)
 *   1       128x128 register-staged GEMM      2       128x256 LDS-DMA ring GEMM
 *   3/4/5   slab kernel, 128/192/256-row tiles   6/7   slab kernel, 32/64-row tiles
 *   200/201/202 slab kernel tile order: plain / XCD-contiguous (default) / XCD-contiguous with column-tile PAIRS per XCD where the launch
 *           has 4, 8 or 16 column tiles and a weight panel larger than an XCD's L2 (the decoder FFN conv1: 109 MB fetched instead of 139,
 *           1 % slower - measured r05); same results
 *   220/221 bf16 pointwise launches of more tiles than CUs: one tile per workgroup / the persistent kernel (default); bit-identical
 *   230/231 wide depth-wise predictors: the last LayerNorm + Linear(filter, 1) head as a normalise pass over stored activations / from
 *           row sums the last GEMM's epilogue leaves (fs2_op_gemm_head, default); equal to fp32 rounding of another summation order
 *   500/501 fp32 slab-kernel launches: fp32 MFMA (default) / bf16 x 3 split products (what FS2_MIXED_X3 uses in its front; operator level)
 *   700/701 bf16 fs2_op_bgemm tile order: plain / XCD-contiguous (default)
 *   800/801 bf16 fs2_op_bgemm: generic instantiation only / the bounds-free one for full, aligned tiles (default)
 *   900/901 fs2_op_attention_bwd, dK / dV launch: one (default, also 909) / two 16-row blocks per wave, 905/906 three / four (one wave
 *              per SIMD; measured no faster inside the training step);  902/903/904 the dQ launch: one / two / by size (default),
 *              907/908 three / four
 *   1000/1001 bf16 fs2_op_bgemm: 256 x 256 LDS-DMA kernel for eligible TN products off / on (default)
 *   1100/1101 fs2_op_col_sum: two launches (default) / one (last workgroup reduces; measured slower)
 *   1200..1204 fused attention: 1200 = the phase-serial kernel only; 1201 / 1202 / 1204 = the software-pipelined kernel (bf16, head dim
 *              128, no attention dropout) with 32 / 64 / 96 queries per wave; 1203 = by size (default)
 *   1400..1402 bf16 GEMMs with K = 256 (no fused epilogue): slab kernel / weight-resident kernel from three row tiles per
 *              workgroup on (default) / wherever it applies; bit-identical outputs
 *   1320/1321  inference engine, bf16: the VarianceEncoder's bucketize + embedding add as the tail of its predictor's launch: off / on (default;
 *              bit-identical either way)
 *   1340/1341  inference engine, bf16, H = 256 with two heads, sequences of up to 768 rows (the encoder stack; the decoder stack for short
 *              utterances): self-attention and out-projection + residual + LayerNorm as two launches / as one (attn_out_ln_kernel, default; r06): the attention rows are the same bits, the out-projection sums its 256
 *              products in another order (equal to fp32 rounding before the bf16 store)
 *   1500/1501  fp32-storage split modes: attention on fp32 MFMA / on bf16 x 3 split products (default)
 * (Removed in r05, measured slower or neutral in r02-r04 and kept until then behind switches: 1211 resident-K/V attention, 211 operand
 *  ring, 1301 / 1302 tall / paired predictor tiles, 301 in-place wide-row LayerNorm epilogue, 311 256-row deferred epilogue.
 *  Removed in r06: 250..253, three one-wave-per-SIMD / operand-ring / producer-consumer forms of the slab GEMM that measured 5-25 % slower
 *  in r05 - their sources and measurements are kept as probes under tools/probes/gemm_forms/, outside the library.) */
int fs2_op_set_gemm_variant(int32_t variant);
/* The two vocoder knobs below are PER CALLING THREAD, like fs2_op_set_gemm_variant: fs2_voc_synthesize reads the switches of the thread
 * that calls it - a value set on the main thread does not reach a vocoder driven from a pipeline / executor worker thread (set it there).
 * tuning knob: cap (KiB) on the LDS operand slab of a vocoder conv workgroup; 0 = built-in heuristic */
int fs2_op_set_vocoder_lds_limit(int32_t kib);
/* A/B knob: 1 (default) = whole resblocks of the 32/64-channel stages as one LDS-resident launch, 0 = conv by conv */
int fs2_op_set_vocoder_fused_resblock(int32_t on);
int fs2_op_gemm(int32_t dtype, int32_t out_dtype, const void* x, const void* w, const float* bias, void* c,
                int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S, int32_t relu, void* hip_stream);
/* c = dropout(relu(x w^T + bias)) with fs2_op_dropout's mask over the (M, N) product in the store (the FFN's hidden tensor in training
 * mode: no read-modify-write pass over the (M, filter) tensor).  FS2_ERR_SHAPE when the shape does not run on the slab kernel. */
int fs2_op_gemm_relu_dropout(int32_t dtype, const void* x, const void* w, const float* bias, void* c, int32_t M, int32_t N, int32_t Cin,
                             int32_t taps, int32_t S, float p, uint64_t seed, uint64_t key, void* hip_stream);
/* c = x w^T + bias + addend, addend (M, N) in the launch dtype; addend == c is allowed (a workgroup reads its own tile of it before it
 * writes it): the accumulating data-gradient products of the training step (dx += dy . W) without an elementwise pass behind them.
 * FS2_ERR_SHAPE when the shape does not run on the slab kernel (N < 192, M % S != 0, even tap count). */
int fs2_op_gemm_add(int32_t dtype, const void* x, const void* w, const float* bias, const void* addend, void* c, int32_t M, int32_t N,
                    int32_t Cin, int32_t taps, int32_t S, void* hip_stream);
/* c = LayerNorm(v) w0^T + bias0 evaluated on the PRE-norm rows v = x (bf16): the caller folds gamma / beta into the operands
 * (w = w0 diag(gamma) as stored in bf16, bias = bias0 + w0 beta, wg[n] = sum_k w[n][k]) and hands over v's finished row
 * statistics rowstats (M) float2 (rstd, rstd * mean) (fs2_op_rowstats_finish); the epilogue applies
 * c[m][n] = rstd[m] * acc - rstd[m] * mean[m] * wg[n] + bias[n].  What the engine's in-projection does behind a deferred norm2
 * (litfass/fastspeech2/model.py:113-115) instead of a normalise-only pass, and the mel Linear behind the last decoder block's
 * (fastspeech2.py:723; N < 192: the 128 x 128 kernel's epilogue).  bf16 only; FS2_ERR_SHAPE otherwise. */
int fs2_op_gemm_rowscale(const void* x, const void* w, const float* bias, const float* rowstats, const float* wg,
                         void* c, int32_t M, int32_t N, int32_t Cin, void* hip_stream);
/* parts (M, nparts) float2 partial (sum, sum of squares) over ncols columns per row - what the deferred-LayerNorm GEMM epilogue
 * leaves, one per 256-column tile - -> out (M) float2 (rstd, rstd * mean) */
int fs2_op_rowstats_finish(const float* parts, int32_t nparts, int32_t ncols, float eps, float* out, int32_t M, void* hip_stream);
/* The last layer of a wide VariancePredictor (litfass/fastspeech2/model.py:512-518,538: ... -> ReLU -> LayerNorm -> Linear(N, 1) ->
 * masked_fill) without storing its activations: v = act(x w^T + bias) (bf16 operands, fp32 v) is reduced in the GEMM epilogue to the
 * row statistics stats_out (M, ceil(N/256)) float2 (sum, sum of squares per 256-column tile) and head_out (M, ceil(N/256)) =
 * sum_n v[m][n] * head_gw[n] per tile, head_gw = gamma * w_head; fs2_op_head_finish then gives
 * pred[m] = mask[m] ? 0 : rstd * (sum of head_out[m][:] - mean * sum_gw) + cst  =  LayerNorm(v)[m] . w_head + b_head with
 * sum_gw = sum_n head_gw[n], cst = beta . w_head + b_head.  bf16, N >= 192, N % 8 == 0, Cin % 128 == 0; FS2_ERR_SHAPE otherwise. */
int fs2_op_gemm_head(const void* x, const void* w, const float* bias, const float* head_gw, float* stats_out, float* head_out, int32_t M,
                     int32_t N, int32_t Cin, int32_t relu, void* hip_stream);
int fs2_op_head_finish(const float* parts, const float* dots, int32_t nparts, int32_t ncols, float eps, float sum_gw, float cst,
                       const uint8_t* mask, float* pred, int32_t M, void* hip_stream);
/* Split-K form for long reductions over few row tiles (training step: the encoder-side data-gradient convs, M = B L rows, K = taps x
 * filter): ksplit (from fs2_op_gemm_splitk_choice; 1 = not worth it -> use fs2_op_gemm) slices of the input channels run as separate
 * workgroups of ONE launch into fp32 planes part (ksplit, M, N), a second launch adds the planes in order:
 * c (out_dtype) = [c +] sum_s part[s].  No bias / ReLU.  FS2_ERR_SHAPE when the shape does not run on the slab kernel. */
int fs2_op_gemm_splitk_choice(int32_t dtype, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S);
int fs2_op_gemm_splitk(int32_t dtype, int32_t out_dtype, const void* x, const void* w, void* c, float* part, int32_t M, int32_t N,
                       int32_t Cin, int32_t taps, int32_t S, int32_t ksplit, int32_t accumulate, void* hip_stream);
/* ... with the ReLU (and dropout) backward of a data-gradient product folded into the store: c = gate > 0 ? scale * (x w^T + bias)
 * : 0, gate (M, N) in the launch dtype (training step: dh = (dc2 . W2) o [h > 0] / (1 - p), h = dropout(relu(.)) of the forward:
 * h > 0 <=> kept and pre-activation > 0, model.py:84-90 backwards).  FS2_ERR_SHAPE when the shape does not run on the slab kernel
 * (N < 192, M % S != 0, even tap count): the caller then uses fs2_op_gemm + fs2_op_dropout + fs2_op_ew(op 1). */
int fs2_op_gemm_gated(int32_t dtype, const void* x, const void* w, const float* bias, const void* gate, float scale, void* c,
                      int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S, void* hip_stream);
int fs2_op_attention(int32_t dtype, const void* qkv, const uint8_t* key_pad_mask, void* out, void* vt_scratch,
                     uint64_t* bits_scratch, int32_t B, int32_t S, int32_t H, int32_t heads, void* hip_stream);
/* The same attention (nn.MultiheadAttention inside ConformerEncoderLayer.forward, litfass/fastspeech2/model.py:108-116) in the
 * split arithmetic of the fp32x3 / mixed3 modes: qkv fp32 (B*S, 3H) is split into bf16 heads and tails (split_scratch: 2 x B*S*3H
 * bf16), every q.k and p.v product is three bf16 MFMAs (lo*hi + hi*lo + hi*hi, fp32 accumulate), softmax fp32; out fp32 (B*S, H). */
int fs2_op_attention_x3(const float* qkv, const uint8_t* key_pad_mask, float* out, void* split_scratch, uint64_t* bits_scratch,
                        int32_t B, int32_t S, int32_t H, int32_t heads, void* hip_stream);
/* fp32 GEMM c = x w^T + bias (split != 0: bf16 x 3 split products) whose result leaves as TWO bf16 (M, N) tensors, c_hi + c_lo = c
 * up to 2^-17 |c| - how the engine's in-projection feeds the split-arithmetic attention without an extra pass.  N >= 192. */
int fs2_op_gemm_split_out(const void* x, const void* w, const float* bias, void* c_hi, void* c_lo, int32_t M, int32_t N, int32_t Cin,
                          int32_t split, void* hip_stream);
/* GEMM/conv with the fused row epilogue  y = LayerNorm(act(xW^T + b) [+ res]) [, pred = head(y)]
 * (y or pred may be NULL; tmp = (M, N) scratch used when the shape cannot be fused) */
int fs2_op_gemm_ln(int32_t dtype, const void* x, const void* w, const float* bias, const void* res,
                   const float* ln_g, const float* ln_b, const float* dot_w, float dot_b, const uint8_t* mask,
                   float* pred, void* y, void* tmp, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S,
                   int32_t relu, void* hip_stream);
/* The same fused launch for the training tape: y = LayerNorm(z) * g + b with z = act(x W^T + bias) [+ res], and z itself stored
 * too (z_out, (M, N), the activation dtype) - what LayerNorm's backward needs (the forward of ConformerEncoderLayer's
 * x = norm(x + sublayer(x)), model.py:114-115, and of a predictor layer's conv -> ReLU -> LayerNorm, model.py:528-538, without a
 * stand-alone LayerNorm launch).  FS2_ERR_SHAPE (nothing launched) where the epilogue does not apply (N > 256, shapes the slab
 * kernel does not take): the caller keeps its GEMM + LayerNorm launches. */
int fs2_op_gemm_ln_tape(int32_t dtype, const void* x, const void* w, const float* bias, const void* res, const float* ln_g,
                        const float* ln_b, void* y, void* z_out, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S,
                        int32_t relu, void* hip_stream);
/* ... with nn.Dropout between the product and the residual add (the residual sites of ConformerEncoderLayer in training mode, model.py:
 * 117-121): z = dropout(act(x w^T + bias)) + res, y = LayerNorm(z); the mask is fs2_op_dropout's over the (M, N) product with this
 * (p, seed, key), so fs2_op_dropout on the gradient regenerates it. */
int fs2_op_gemm_ln_tape_dropout(int32_t dtype, const void* x, const void* w, const float* bias, const void* res, const float* ln_g,
                                const float* ln_b, void* y, void* z_out, int32_t M, int32_t N, int32_t Cin, int32_t taps, int32_t S,
                                int32_t relu, float p, uint64_t seed, uint64_t key, void* hip_stream);
size_t fs2_op_attention_scratch_bytes(int32_t dtype, int32_t B, int32_t S, int32_t H, int32_t heads, size_t* bits_bytes);
int fs2_op_layernorm(int32_t dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                     const float* dot_w, float dot_b, const uint8_t* mask, float* pred, int32_t M, int32_t H,
                     void* hip_stream);
int fs2_op_dwconv(int32_t dtype, const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t S,
                  int32_t C, int32_t k, void* hip_stream);
int fs2_op_durations(const float* dur_pred, const uint8_t* src_mask, const int32_t* forced, int32_t* dur,
                     int32_t* cum, int32_t* totals, int32_t* guard, int32_t B, int32_t L, void* hip_stream);
int fs2_op_regulate(int32_t dtype, const void* x, const int32_t* cum, const int32_t* totals, void* y,
                    uint8_t* tgt_mask, int32_t B, int32_t L, int32_t T, int32_t H, void* hip_stream);
int fs2_op_bucket_embed(int32_t dtype, const void* x, const float* pred, const float* bins, const float* emb,
                        int32_t nbins, float std, float mean, const float* pe, const float* spk, void* y,
                        int32_t* idx_out, int32_t B, int32_t T, int32_t H, void* hip_stream);
int fs2_op_embed(int32_t dtype, const int64_t* phones, const float* table, const float* pe, const float* spk,
                 void* x, uint8_t* src_mask, int32_t B, int32_t L, int32_t H, int32_t n_phones, void* hip_stream);
int fs2_op_spk_proj(const float* dvec, const float* w, const float* b, float* spk, int32_t B, int32_t H,
                    int32_t Din, void* hip_stream);
/* The encoder-side fused launch (r06; nn.MultiheadAttention's core + out_proj + the residual + norm1 of ConformerEncoderLayer.forward,
 * model.py:108-116, behind the in-projection GEMM): out = LayerNorm(res + softmax(q k^T / sqrt(d) + key padding) v w_out^T + bias), bf16,
 * H = 256, two heads.  qkv (B*S, 3H), key_pad_mask (B, S) 1 = pad, w_out (H, H) bf16, res / out (B*S, H) bf16 (out may alias res),
 * scratch = H * H * 2 + B * ceil(S / 64) * 8 bytes.  FS2_ERR_SHAPE for other shapes / dtypes. */
int fs2_op_attn_out_ln(int32_t dtype, const void* qkv, const uint8_t* key_pad_mask, const void* w_out, const float* bias, const void* res,
                       const float* ln_g, const float* ln_b, void* out, void* scratch, int32_t B, int32_t S, int32_t H, int32_t heads,
                       void* hip_stream);
/* VariancePredictor (model.py:482-522), dense k=3, H=256, bf16, one launch: w = (nlayers, H, taps*H)
 * tap-major rows, bias / ln_g / ln_b = (nlayers, H) fp32, packed_scratch = nlayers * H * taps * H * 2
 * bytes (fragment-ordered copy of w, built by this call).  FS2_ERR_SHAPE if the shape is not covered. */
int fs2_op_predictor(int32_t dtype, const void* x, const void* w, const float* bias, const float* ln_g,
                     const float* ln_b, const float* head_w, float head_b, const uint8_t* mask, float* pred,
                     void* packed_scratch, int32_t B, int32_t S, int32_t H, int32_t nlayers, int32_t taps,
                     void* hip_stream);
/* The same for DEPTH-WISE layers (VarianceConvolutionLayer with depthwise=True, model.py:541-558: Conv1d(H, H, 3, groups = H) ->
 * Conv1d(H, H, 1) -> ReLU -> LayerNorm), H = 256, bf16, one launch: dw_w = (nlayers, 3, H) fp32 tap-major depth-wise taps, dw_b =
 * (nlayers, H), w = (nlayers, H, H) bf16 pointwise weights, bias their bias; packed_scratch = nlayers * H * H * 2 bytes. */
int fs2_op_predictor_dw(int32_t dtype, const void* x, const float* dw_w, const float* dw_b, const void* w, const float* bias,
                        const float* ln_g, const float* ln_b, const float* head_w, float head_b, const uint8_t* mask, float* pred,
                        void* packed_scratch, int32_t B, int32_t S, int32_t H, int32_t nlayers, void* hip_stream);
/* FastSpeech2Loss.get_loss for "l1" / "mse" (litfass/fastspeech2/loss.py:57-81, called from forward :83-213):
 * out2[0] = mean over the rows whose pad_mask is 0 of |pred - truth| (kind 0) or (pred - truth)^2 (kind 1),
 * out2[1] = number of selected elements; pred = (rows, inner) fp32; truth_kind 0: fp32 (rows, inner),
 * 1: int64 target durations compared as log(d + 1) (loss.py:176); ws = fs2_op_masked_loss_ws_bytes() device
 * bytes, zero-filled once by the caller and reusable across calls on one stream.  Deterministic (fp64
 * partials added in a fixed order, no float atomics).  All device pointers. */
size_t fs2_op_masked_loss_ws_bytes(void);
int fs2_op_masked_loss(const float* pred, const void* truth, int32_t truth_kind, const uint8_t* pad_mask, int64_t rows,
                       int32_t inner, int32_t kind, void* ws, float* out2, void* hip_stream);
/* Soft-DTW value of B sequence pairs (litfass/third_party/softdtw/__init__.py:8-24,110-117 = the validation metric of
 * fastspeech2.py:1149-1156; the same recursion is the arithmetic of the "soft_dtw" loss kind, loss.py:57-81):
 * out[b] = R[N, M] with D[i,j] = |x[b,i] - y[b,j]|^2 (fp32), the recursion in fp64.  x (B, N, D), y (B, M, D), out (B): fp32
 * device pointers.  One workgroup per pair; the sequences are staged in LDS when both fit beside the three fp64
 * anti-diagonals (e.g. 2 x 240 frames of an 80-bin mel), else read through the caches; FS2_ERR_SHAPE for N > 6800. */
int fs2_op_soft_dtw(const float* x, const float* y, int32_t B, int32_t N, int32_t M, int32_t D, float gamma, float* out,
                    void* hip_stream);
/* The same value plus its gradient with respect to x - what loss.backward() yields through the reference's vendored module
 * (third_party/softdtw/__init__.py:27-52 compute_softdtw_backward on the float32 R / D that _SoftDTW.forward saves, :56-77;
 * autograd through calc_distance_matrix :85-92): grad_x[b,i,:] = 2 sum_j E[b,i,j] (x[b,i,:] - y[b,j,:]).  The "soft_dtw" loss
 * kind of loss.py:36,62-81 differentiates exactly this, chunk by chunk, with respect to the prediction only.  scratch: device
 * memory of fs2_op_soft_dtw_grad_scratch_bytes(B, N, M) bytes (R, D and E of every pair).  Deterministic. */
size_t fs2_op_soft_dtw_grad_scratch_bytes(int32_t B, int32_t N, int32_t M);
int fs2_op_soft_dtw_grad(const float* x, const float* y, int32_t B, int32_t N, int32_t M, int32_t D, float gamma, float* out,
                         float* grad_x, void* scratch, size_t scratch_bytes, void* hip_stream);
/* dtype conversion helpers for tests: fp32 <-> engine dtype, n elements, device pointers */
int fs2_op_convert(int32_t src_dtype, int32_t dst_dtype, const void* src, void* dst, size_t n, void* hip_stream);

/* ================================================================================================
 * Training step: the operators the reference gets from autograd in
 * FastSpeech2.training_step (litfass/fastspeech2/fastspeech2.py:786-797: loss.backward()) and from
 * torch.optim.AdamW + gradient_clip_val (fastspeech2.py:1166-1182, scripts/train.sh:16).  The tape itself
 * (which tensor feeds which gradient) lives in the host mirror, lightningfastspeech2_amd/training.py.
 * All device pointers; fp32 (FS2_F32) arithmetic on the exact fp32 MFMA.
 * ================================================================================================ */
/* C[b1][b2](m, n) = alpha * sum_k A(m, k) B(k, n) + bias[n] + beta * C, operands by element strides (one stride of each
 * operand is 1), two batch levels, optional deterministic split-K, and the two implicit 'same'-padded Conv1d forms over
 * (B*S, C) time-major rows with utterances of `seg` rows:
 *   data gradient (taps > 1):  K = taps * Kin; the k-tiles of tap j read A rows m + a_shift0 + j * a_shift_step
 *                              (zero outside the row's utterance) and B from + j * sBtap;
 *   weight gradient (taps <= 1, seg > 0): batch index b2 reads B's k rows at k + b_shift0 + b2 * b_shift_step. */
typedef struct fs2_bgemm_desc {
    int32_t M, N, K;
    int64_t sAm, sAk, sBk, sBn, ldc;
    int32_t nb1, nb2;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha, beta;
    int32_t splitk;
    int32_t seg, taps, Kin, a_shift0, a_shift_step;
    int64_t sBtap;
    int32_t b_shift0, b_shift_step;
    int32_t c_dtype;   /* fs2_dtype of C: FS2_F32 operands write fp32; FS2_BF16 operands write bf16 or fp32 */
} fs2_bgemm_desc;
size_t fs2_op_bgemm_ws_bytes(const fs2_bgemm_desc* d);  /* split-K slabs (0 when splitk <= 1) */
/* 1 when bf16 operands of this shape run on the 256 x 256 tile kernel (TN product, M % 256 == N % 256 == K % 32 == 0, fp32 C):
 * the caller sizes splitk for 4x fewer, 512-thread workgroups (one per CU) then */
int32_t fs2_op_bgemm_tn256(const fs2_bgemm_desc* d);
int fs2_op_bgemm(int32_t dtype, const fs2_bgemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                 float* ws, void* hip_stream);
/* The fused (flash) attention of the inference path on the training path: also writes lse2 (B, heads, S) - per query, log2 of
 * the softmax denominator in the scaled scores' log2 units - and applies nn.MultiheadAttention's attention-weight dropout
 * (drop_p = 0: none; mask regenerated from (seed, key) over the (b, head, query, key) index); and its recomputing backward
 * (bf16, head dim 128: fs2_op_attention_bwd_supported): dqkv (B*S, 3H) from dout (B*S, H), qkv, lse2 and
 * delta = fs2_op_attn_delta(dout, out); neither the probabilities nor their gradients reach HBM.  Two launches, no atomics. */
int fs2_op_attention_train(int32_t dtype, const void* qkv, const uint8_t* key_pad_mask, void* out, void* vt_scratch,
                           uint64_t* bits_scratch, float* lse2, int32_t B, int32_t S, int32_t H, int32_t heads, float drop_p,
                           uint64_t drop_seed, uint64_t drop_key, void* hip_stream);
int32_t fs2_op_attention_bwd_supported(int32_t dtype, int32_t H, int32_t heads);
int fs2_op_attention_bwd(int32_t dtype, const void* qkv, const void* dout, const float* lse2, const float* delta,
                         const uint8_t* key_pad_mask, void* dqkv, int32_t B, int32_t S, int32_t H, int32_t heads, float drop_p,
                         uint64_t drop_seed, uint64_t drop_key, void* hip_stream);
/* attention backward without a dP tensor: C = dS = alpha * P o (dropout(A B) - delta) where A B = dO V^T is the product the
 * descriptor states, P (C's layout and dtype) are the forward's probabilities, delta (nb1, nb2, M) = fs2_op_attn_delta(dO, O)
 * (= sum_k dP P) and the dropout is the forward's attention-weight dropout (drop_p = 0: none), regenerated from (seed, key) */
int fs2_op_bgemm_softmax_bwd(int32_t dtype, const fs2_bgemm_desc* d, const void* A, const void* B, void* C, const void* P,
                             const float* delta, float drop_p, uint64_t drop_seed, uint64_t drop_key, void* hip_stream);
int fs2_op_attn_delta(int32_t dtype, const void* dout, const void* out, float* delta, int32_t B, int32_t S, int32_t H,
                      int32_t heads, void* hip_stream);
/* LayerNorm backward of y = LN(z [+ res]) * gamma + beta: dz (M, H); relu_mask = 1 when z is a ReLU output and dz should be
 * the gradient of the pre-activation (dz zeroed where z <= 0).  part = (fs2_op_layernorm_bwd_parts(M), 3, H) partial column
 * sums of dy * zhat, dy and dz, to be reduced with fs2_op_col_sum over the parts -> dgamma, dbeta and the bias gradient of
 * the layer that produced z.  H % 4 == 0, H <= 1024. */
int32_t fs2_op_layernorm_bwd_parts(int32_t M);
int fs2_op_layernorm_bwd(int32_t dtype, const void* z, const void* res, const void* dy, const float* gamma, void* dz,
                         float* part, int32_t M, int32_t H, int32_t relu_mask, void* hip_stream);
/* ... whose dy is the gradient of dropout(y) (VarianceConvolutionLayer: LayerNorm -> Dropout, model.py:538-539,556-557): the
 * forward's mask (fs2_op_dropout / fs2_op_layernorm_dropout with the same seed, key) is applied to dy on load */
int fs2_op_layernorm_bwd_dropout(int32_t dtype, const void* z, const void* res, const void* dy, const float* gamma, void* dz,
                                 float* part, int32_t M, int32_t H, int32_t relu_mask, float drop_p, uint64_t seed, uint64_t key,
                                 void* hip_stream);
/* ... for y = LayerNorm(res + dropout(u)) (the residual sites): dz as above (the residual's gradient) AND dzm = mask o dz / (1 - out_p),
 * the gradient of u, as a second tensor (mask of fs2_op_dropout(., out_p, seed, out_key) over (M, H)); the third column sum of part is
 * then dzm's (u's bias gradient). */
int fs2_op_layernorm_bwd_masked(int32_t dtype, const void* z, const void* res, const void* dy, const float* gamma, void* dz, void* dzm,
                                float* part, int32_t M, int32_t H, int32_t relu_mask, float out_p, uint64_t seed, uint64_t out_key,
                                void* hip_stream);
/* fs2_op_layernorm with its fused Linear(H, 1) head reading the bias from DEVICE memory (a parameter being trained: the host
 * copy would cost a read-back + stream sync per step) */
int fs2_op_layernorm_head(int32_t dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                          const float* dot_w, const float* dot_b_dev, const uint8_t* mask, float* pred, int32_t M, int32_t H,
                          void* hip_stream);
/* y = dropout(LayerNorm(x [+ res])) in one launch; the mask is fs2_op_dropout's over the (M, H) element index */
int fs2_op_layernorm_dropout(int32_t dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                             int32_t M, int32_t H, float drop_p, uint64_t seed, uint64_t key, void* hip_stream);
/* out[s][n] (+)= scale * sum over the rows of segment s of x[row][n]; seg = rows per segment (0: one segment).  Per-chunk
 * partials in ws (fs2_op_col_sum_ws_bytes), added in chunk order (deterministic).  ws starts with 8192 32-bit counters for the
 * one-launch variant (fs2_op_set_gemm_variant 1101; off by default, measured slower): the caller zeroes them ONCE, when it
 * allocates ws; every launch leaves them zero again.
 * fs2_op_col_sum2: one segment, columns [0, n1) to out and [n1, N) to out2, each with its own accumulate flag. */
size_t fs2_op_col_sum_ws_bytes(int32_t M, int32_t N, int32_t seg);
int fs2_op_col_sum(int32_t dtype, const void* x, float* out, float* ws, int32_t M, int32_t N, int32_t ldx, int32_t seg,
                   int32_t accumulate, float scale, void* hip_stream);
int fs2_op_col_sum2(int32_t dtype, const void* x, float* out, float* out2, int32_t n1, float* ws, int32_t M, int32_t N,
                    int32_t ldx, int32_t accumulate, int32_t accumulate2, float scale, void* hip_stream);
/* out[n] (+)= scale * sum_r row_w[r] * x[r][n]: the weight gradient of a Linear(H, 1) head (row_w = the loss gradient per row) */
int fs2_op_col_sum_weighted(int32_t dtype, const void* x, const float* row_w, float* out, float* ws, int32_t M, int32_t N,
                            int32_t ldx, int32_t accumulate, float scale, void* hip_stream);
/* masked softmax over the key axis of (B, heads, S, S) fp32 scores -> probabilities p in the activation dtype (the training
 * path materialises them; p may alias s for FS2_F32), and its backward ds = scale * P o (dP - sum_k dP o P) from fp32 dP */
int fs2_op_softmax_fwd(int32_t dtype, const float* s, const uint8_t* key_pad, void* p, int32_t B, int32_t heads, int32_t S,
                       float scale, void* hip_stream);
int fs2_op_softmax_bwd(int32_t dtype, const float* dp, const void* p, void* ds, int32_t B, int32_t heads, int32_t S,
                       float scale, void* hip_stream);
/* elementwise over the activation dtype: op 0: out = alpha*a + beta*b (b may be NULL)   1: out = a where b > 0 else 0
 * (ReLU backward)   2: out = alpha*a */
int fs2_op_ew(int32_t dtype, int32_t op, const void* a, const void* b, void* out, size_t n, float alpha, float beta,
              void* hip_stream);
/* embedding backward: table[idx[r]] += x[r] for r < R (int32 or int64 indices), row skip_row untouched (padding_idx).
 * ws: fs2_op_scatter_rows_ws_bytes(R, H, V) bytes when that is non-zero (long index lists over tables of <= 256 rows take a
 * chunked two-phase form; ws follows fs2_op_col_sum's zero-once rule), else NULL; NULL always selects the one-launch kernel.
 * Either form is deterministic. */
size_t fs2_op_scatter_rows_ws_bytes(int32_t R, int32_t H, int32_t V);
int fs2_op_scatter_rows(int32_t dtype, const void* x, const int32_t* idx32, const int64_t* idx64, float* table, float* ws,
                        int32_t R, int32_t H, int32_t V, int32_t skip_row, void* hip_stream);
/* LengthRegulator backward: dx[b][p] = sum of dy[b][t] over the frames phone p was repeated to (truncated at T) */
int fs2_op_regulate_bwd(int32_t dtype, const void* dy, const int32_t* cum, void* dx, int32_t B, int32_t L, int32_t T,
                        int32_t H, void* hip_stream);
/* gradient of alpha * fs2_op_masked_loss(...) with respect to pred; stat = that call's out2 (the count is read on device) */
int fs2_op_masked_loss_bwd(const float* pred, const void* truth, int32_t truth_kind, const uint8_t* pad_mask,
                           const float* stat, float* dpred, int64_t rows, int32_t inner, int32_t kind, float alpha,
                           void* hip_stream);
/* PriorEmbedding (model.py:146-164) on the training path: y[b, t] = x[b, t] + emb[bucketize(values[b], bins)] for every row of
 * utterance b; emb is the caller's relu(embedding.weight) (relu(E[i]) == relu(E)[i]); idx_out (B*T) receives the bucket per row */
int fs2_op_bucket_embed_utt(int32_t dtype, const void* x, const float* values, const float* bins, const float* emb, int32_t nbins,
                            void* y, int32_t* idx_out, int32_t B, int32_t T, int32_t H, void* hip_stream);
/* nn.Dropout in training mode: y = x * keep / (1 - p) (y may alias x).  The mask is a counter-based hash of (seed, key, element
 * index) - nothing is stored; the backward applies the same call to the gradient.  Not the reference's random stream (torch's
 * Philox), the same distribution. */
int fs2_op_dropout(int32_t dtype, const void* x, void* y, size_t n, float p, uint64_t seed, uint64_t key, void* hip_stream);
/* pred[m] = mask[m] ? 0 : y[m] . w + b[0]: the VariancePredictor head (model.py:512-518) when a dropout layer sits between the
 * last LayerNorm and the Linear (otherwise fs2_op_layernorm's fused head does it) */
int fs2_op_row_dot(int32_t dtype, const void* y, const float* w, const float* b, const uint8_t* mask, float* pred, int64_t M,
                   int32_t H, void* hip_stream);
/* depth-wise Conv1d (model.py:75-81, 545-551) backward: data gradient = the same conv with the taps reversed and no bias;
 * weight / bias gradient as per-chunk partials part (fs2_op_dwconv_wgrad_parts(B, S), C * (k + 1)): [c * k + j] then [C * k + c],
 * reduced with fs2_op_col_sum into the adjacent (C, k) weight and (C) bias gradients */
int fs2_op_dwconv_dgrad(int32_t dtype, const void* dy, const float* w, void* dx, int32_t B, int32_t S, int32_t C, int32_t k,
                        void* hip_stream);
int32_t fs2_op_dwconv_wgrad_parts(int32_t B, int32_t S);
int fs2_op_dwconv_wgrad(int32_t dtype, const void* dy, const void* x, float* part, int32_t B, int32_t S, int32_t C, int32_t k,
                        void* hip_stream);
/* the depth-wise layer's conv2 = Sequential(grouped 1x1 conv, groups = H over F channels; pointwise F -> H) (model.py:84-93)
 * as one linear map Wf (H, F) = W21 blockdiag(G), bf = b21 + W21 bg (what the inference engine folds once at load time), and
 * the chain rule from (dWf, dbf) back to the four parameter gradients (accumulated) */
int fs2_op_fold_conv2(int32_t wf_dtype, const float* G, const float* bg, const float* W21, const float* b21, void* Wf, float* bf,
                      int32_t H, int32_t F, void* hip_stream);
int fs2_op_unfold_conv2(const float* dWf, const float* dbf, const float* G, const float* bg, const float* W21, float* dG,
                        float* dbg, float* dW21, float* db21, int32_t H, int32_t F, void* hip_stream);
/* the data-gradient kernel of a Conv1d / Linear: dst (Cin, taps*N), dst[ci][j'*N + n] = src[n][(taps-1-j')*Cin + ci] for
 * src (N, taps*Cin) tap-major, so that dX = fs2_op_gemm(x = dY, w = dst, M, N = Cin, Cin = N, taps, S) */
int fs2_op_transpose_weight(int32_t dtype, const void* src, void* dst, int32_t N, int32_t Cin, int32_t taps, void* hip_stream);
/* ... for many bf16 weights in one launch: table_dev = n rows of 6 int64 ON THE DEVICE, {src, dst, N, Cin, taps, first tile},
 * a weight owning fs2_op_transpose_weight_tiles(N, Cin, taps) consecutive tiles from its first tile; tiles = their total */
int64_t fs2_op_transpose_weight_tiles(int32_t N, int32_t Cin, int32_t taps);
int fs2_op_transpose_weight_batch(const int64_t* table_dev, int32_t n, int64_t tiles, void* hip_stream);
/* out[0] = sum of squares of x (fp64 partials, fixed order) */
size_t fs2_op_sum_sq_ws_bytes(size_t n);
int fs2_op_sum_sq(const float* x, size_t n, float* ws, float* out, void* hip_stream);
/* torch.optim.AdamW step over flat fp32 buffers; g is scaled by grad_scale and, when gnorm_sq (device scalar, sum of
 * squares of g) is given, by the clip_grad_norm_ coefficient min(1, max_norm / (grad_scale * sqrt(gnorm_sq) + 1e-6)) */
int fs2_op_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale,
                 void* hip_stream);
/* ... that also writes the updated weights rounded to bf16 (shadow_bf16, n elements: the mixed-precision path's operands) */
int fs2_op_adamw_shadow(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int32_t step, const float* gnorm_sq, float max_norm,
                        float grad_scale, void* hip_stream);
/* teacher-forced VarianceEncoder embedding (model.py:417-422): y = x + Emb[bucketize(target * std + mean)] (+ pe + spk) */
int fs2_op_bucket_embed_target(int32_t dtype, const void* x, const float* target, const float* bins, const float* emb,
                               int32_t nbins, float std, float mean, const float* pe, const float* spk, void* y,
                               int32_t* idx_out, int32_t B, int32_t T, int32_t H, void* hip_stream);

/* ================================================================================================
 * HiFi-GAN generator: the step right after the mel forward.  Replaces
 * litfass.third_party.hifigan.Synthesiser.__call__ -> Generator.forward
 * (litfass/third_party/hifigan/__init__.py:19-43, models.py:112-165), resblock type "1".
 * Weights are passed under the generator's own state_dict names AFTER remove_weight_norm
 * ("conv_pre.weight", "ups.0.weight" (Cin, Cout, k), "resblocks.3.convs1.0.weight", "conv_post.bias" ...);
 * the host mirror (lightningfastspeech2_amd/hifigan.py) folds weight_g / weight_v pairs.
 * ================================================================================================ */
#define FS2_VOC_MAX_STAGES 8
#define FS2_VOC_MAX_KERNELS 4
typedef struct fs2_voc_config {
    int32_t abi_version;      /* FS2_ABI_VERSION */
    int32_t dtype;            /* fs2_dtype */
    int32_t n_mels;           /* 80 */
    int32_t initial_channel;  /* upsample_initial_channel (config.json:13); halves per stage, multiples of 32 */
    int32_t n_stages;
    int32_t up_rates[FS2_VOC_MAX_STAGES];     /* upsample_rates; kernel - 2 * padding must equal the rate */
    int32_t up_kernels[FS2_VOC_MAX_STAGES];   /* upsample_kernel_sizes */
    int32_t n_kernels;                        /* len(resblock_kernel_sizes) */
    int32_t rb_kernels[FS2_VOC_MAX_KERNELS];  /* odd */
    int32_t rb_dilations[FS2_VOC_MAX_KERNELS][3];
} fs2_voc_config;
typedef struct fs2_vocoder fs2_vocoder;

int fs2_voc_create(const fs2_voc_config* cfg, fs2_vocoder** out);
void fs2_voc_destroy(fs2_vocoder* v);
const char* fs2_voc_last_error(const fs2_vocoder* v);
/* host fp32 data; ndim/shape as torch reports them */
int fs2_voc_load_weight(fs2_vocoder* v, const char* name, const float* data, const int64_t* shape, int32_t ndim);
int fs2_voc_finalize(fs2_vocoder* v);
/* samples per mel frame = prod(up_rates) */
int32_t fs2_voc_hop(const fs2_vocoder* v);
/* mel: (B, T, n_mels) fp32 device (FastSpeech2's output layout); lengths: (B) int32 device valid frames
 * per utterance or NULL; wav: (B, T * hop) fp32 device in [-1, 1] (samples past an utterance's length
 * are left untouched).  Each utterance is synthesised as the reference does it: alone, zero padding at
 * its own ends in every layer (generator.py:163-170). */
int fs2_voc_synthesize(fs2_vocoder* v, const float* mel, const int32_t* lengths, int32_t B, int32_t T, float* wav,
                       void* hip_stream);
/* parity tap: copy stage output `stage` (0 = conv_pre, i = after upsample stage i) of the last call as
 * fp32 (B, T * up_i, C_i) into a device buffer */
int fs2_voc_debug_copy(fs2_vocoder* v, int32_t stage, float* dst, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FS2_H_ */
