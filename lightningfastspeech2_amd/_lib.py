"""ctypes binding of libfs2_hip.so (C ABI declared in include/fs2.h).

The product path has no CPU fallback: if the HIP library is missing or does not export the full
ABI, importing the engine raises.  ``torch`` is imported first on purpose — it loads the ROCm
runtime (libamdhip64.so.7) this library then shares, so torch device pointers and streams are
valid in our launches.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FS2_LIB: alternative build of the same ABI (kernel A/B experiments); default = the in-tree library
LIB_PATH = os.environ.get("FS2_LIB") or os.path.join(_HERE, "libfs2_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fs2.h")

FS2_ABI_VERSION = 4
FS2_MAX_LAYERS = 32
FS2_MAX_VARIANCES = 4
FS2_MAX_PRIORS = 8
FS2_NAME_LEN = 32
FS2_OK = 0
FS2_ERR_HIP, FS2_ERR_SHAPE, FS2_ERR_ARG, FS2_ERR_WEIGHT, FS2_ERR_STATE, FS2_ERR_NOMEM = 1, 2, 3, 4, 5, 6
FS2_F32, FS2_BF16, FS2_MIXED, FS2_MIXED_X3, FS2_F32_X3 = 0, 1, 2, 3, 4
K_CONV_GEMM, K_GEMM, K_ATTENTION, K_ROWOPS, K_DEC_FFN_CONV1, K_DEC_ATTENTION, K_ENC_MHA, K_PREDICTOR = 0, 1, 2, 3, 4, 5, 6, 7


class Fs2ConfigC(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("dtype", C.c_int32), ("n_phones", C.c_int32), ("hidden", C.c_int32),
        ("n_mels", C.c_int32), ("dvec_dim", C.c_int32), ("max_frames", C.c_int32), ("pe_len", C.c_int32),
        ("enc_layers", C.c_int32), ("enc_heads", C.c_int32), ("enc_filter", C.c_int32), ("enc_depthwise", C.c_int32),
        ("enc_kernels", C.c_int32 * FS2_MAX_LAYERS),
        ("dec_layers", C.c_int32), ("dec_heads", C.c_int32), ("dec_filter", C.c_int32), ("dec_depthwise", C.c_int32),
        ("dec_kernels", C.c_int32 * FS2_MAX_LAYERS),
        ("n_variances", C.c_int32),
        ("var_names", (C.c_char * FS2_NAME_LEN) * FS2_MAX_VARIANCES),
        ("var_nlayers", C.c_int32 * FS2_MAX_VARIANCES),
        ("var_kernel", C.c_int32 * FS2_MAX_VARIANCES),
        ("var_mean", C.c_float * FS2_MAX_VARIANCES),
        ("var_std", C.c_float * FS2_MAX_VARIANCES),
        ("var_filter", C.c_int32), ("var_nbins", C.c_int32), ("var_depthwise", C.c_int32),
        ("dur_nlayers", C.c_int32), ("dur_kernel", C.c_int32), ("dur_filter", C.c_int32), ("dur_depthwise", C.c_int32),
        ("n_priors", C.c_int32),
        ("prior_names", (C.c_char * FS2_NAME_LEN) * FS2_MAX_PRIORS),
        ("var_cwt", C.c_int32 * FS2_MAX_VARIANCES),
        ("var_level", C.c_int32 * FS2_MAX_VARIANCES),
    ]


class Fs2OutputsC(C.Structure):
    _fields_ = [
        ("mel", C.c_void_p), ("duration_prediction", C.c_void_p), ("duration_rounded", C.c_void_p),
        ("src_mask", C.c_void_p), ("tgt_mask", C.c_void_p), ("variances", C.c_void_p * FS2_MAX_VARIANCES),
        ("var_spectrogram", C.c_void_p * FS2_MAX_VARIANCES), ("var_mean_std", C.c_void_p * FS2_MAX_VARIANCES),
    ]


class BGemmDescC(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("sAm", C.c_int64), ("sAk", C.c_int64), ("sBk", C.c_int64), ("sBn", C.c_int64), ("ldc", C.c_int64),
        ("nb1", C.c_int32), ("nb2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64), ("sC1", C.c_int64), ("sC2", C.c_int64),
        ("alpha", C.c_float), ("beta", C.c_float), ("splitk", C.c_int32),
        ("seg", C.c_int32), ("taps", C.c_int32), ("Kin", C.c_int32), ("a_shift0", C.c_int32), ("a_shift_step", C.c_int32),
        ("sBtap", C.c_int64), ("b_shift0", C.c_int32), ("b_shift_step", C.c_int32), ("c_dtype", C.c_int32),
    ]


def declared_symbols(header_path: str = HEADER_PATH):
    """Every function include/fs2.h declares (used to verify the library exports the full ABI)."""
    with open(header_path) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs2_[a-z0-9_]+)\s*\(", text)))


class Fs2LibraryError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libfs2_hip.so and type its entry points.  Raises Fs2LibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Fs2LibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise Fs2LibraryError(f"{LIB_PATH} does not export: {missing}")
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    lib.fs2_abi_version.restype = C.c_int
    lib.fs2_status_string.restype = C.c_char_p
    lib.fs2_status_string.argtypes = [C.c_int]
    lib.fs2_last_error.restype = C.c_char_p
    lib.fs2_last_error.argtypes = [vp]
    lib.fs2_create.argtypes = [C.POINTER(Fs2ConfigC), C.POINTER(vp)]
    lib.fs2_destroy.argtypes = [vp]
    lib.fs2_clone.argtypes = [vp, C.POINTER(vp)]
    lib.fs2_load_weight.argtypes = [vp, C.c_char_p, vp, i64p, i32]
    lib.fs2_finalize.argtypes = [vp]
    lib.fs2_encode.argtypes = [vp, vp, vp, i32, i32, vp, vp, C.POINTER(i32)]
    lib.fs2_last_totals.argtypes = [vp, vp, vp, i32]
    lib.fs2_set_priors.argtypes = [vp, vp, i32]
    lib.fs2_decode.argtypes = [vp, C.POINTER(Fs2OutputsC), vp]
    lib.fs2_set_debug.argtypes = [vp, i32]
    lib.fs2_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.fs2_set_workspace.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    lib.fs2_set_frames.argtypes = [vp, i32]
    lib.fs2_set_zero_pad_mel.argtypes = [vp, i32]
    lib.fs2_set_fused_predictor.argtypes = [vp, i32]
    lib.fs2_set_graphs.argtypes = [vp, i32]
    lib.fs2_graph_replays.restype = C.c_int64
    lib.fs2_graph_replays.argtypes = [vp]
    lib.fs2_set_deferred_layernorm.argtypes = [vp, i32]
    lib.fs2_set_folded_layernorm.argtypes = [vp, i32]
    lib.fs2_set_tuning.argtypes = [vp, i32]
    lib.fs2_debug_copy.argtypes = [vp, C.c_char_p, vp, vp]
    lib.fs2_force_buckets.argtypes = [vp, i32, vp]
    lib.fs2_force_variance_targets.argtypes = [vp, i32, vp]
    lib.fs2_profile_enable.argtypes = [vp, i32, i32]
    lib.fs2_profile_reserve.argtypes = [vp, i32, i32]
    lib.fs2_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.fs2_op_convert.argtypes = [i32, i32, vp, vp, C.c_size_t, vp]
    lib.fs2_op_predictor.argtypes = [i32, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_attn_out_ln.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_predictor_dw.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_masked_loss_ws_bytes.restype = C.c_size_t
    lib.fs2_op_masked_loss_ws_bytes.argtypes = []
    lib.fs2_op_masked_loss.argtypes = [vp, vp, i32, vp, C.c_int64, i32, i32, vp, vp, vp]
    lib.fs2_op_soft_dtw.argtypes = [vp, vp, i32, i32, i32, i32, C.c_float, vp, vp]
    lib.fs2_op_soft_dtw_grad_scratch_bytes.argtypes = [i32, i32, i32]
    lib.fs2_op_soft_dtw_grad_scratch_bytes.restype = C.c_size_t
    lib.fs2_op_soft_dtw_grad.argtypes = [vp, vp, i32, i32, i32, i32, C.c_float, vp, vp, vp, C.c_size_t, vp]
    lib.fs2_op_set_gemm_variant.argtypes = [i32]
    lib.fs2_op_set_vocoder_lds_limit.argtypes = [i32]
    lib.fs2_op_set_vocoder_fused_resblock.argtypes = [i32]
    lib.fs2_op_gemm.argtypes = [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_gemm_relu_dropout.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.c_float, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_gemm_add.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_gemm_rowscale.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fs2_op_rowstats_finish.argtypes = [vp, i32, i32, C.c_float, vp, i32, vp]
    lib.fs2_op_gemm_head.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_head_finish.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, vp, vp, i32, vp]
    lib.fs2_op_gemm_splitk_choice.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.fs2_op_gemm_splitk.argtypes = [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_gemm_gated.argtypes = [i32, vp, vp, vp, vp, C.c_float, vp, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_gemm_ln_tape.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_gemm_ln_tape_dropout.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.c_float, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_gemm_ln.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.fs2_op_attention_scratch_bytes.restype = C.c_size_t
    lib.fs2_op_attention_scratch_bytes.argtypes = [i32, i32, i32, i32, i32, C.POINTER(C.c_size_t)]
    lib.fs2_op_attention.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_attention_x3.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_gemm_split_out.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_layernorm.argtypes = [i32, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, i32, i32, vp]
    lib.fs2_op_dwconv.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_durations.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.fs2_op_regulate.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_bucket_embed.argtypes = [i32, vp, vp, vp, vp, i32, C.c_float, C.c_float, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fs2_op_embed.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_spk_proj.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    f32, sz = C.c_float, C.c_size_t
    lib.fs2_op_bgemm_ws_bytes.restype = sz
    lib.fs2_op_bgemm_ws_bytes.argtypes = [C.POINTER(BGemmDescC)]
    lib.fs2_op_bgemm_tn256.restype = i32
    lib.fs2_op_bgemm_tn256.argtypes = [C.POINTER(BGemmDescC)]
    lib.fs2_op_bgemm.argtypes = [i32, C.POINTER(BGemmDescC), vp, vp, vp, vp, vp, vp]
    lib.fs2_op_attention_train.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_attention_bwd_supported.restype = i32
    lib.fs2_op_attention_bwd_supported.argtypes = [i32, i32, i32]
    lib.fs2_op_attention_bwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_bgemm_softmax_bwd.argtypes = [i32, C.POINTER(BGemmDescC), vp, vp, vp, vp, vp, f32, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_attn_delta.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_layernorm_bwd_parts.restype = i32
    lib.fs2_op_layernorm_bwd_parts.argtypes = [i32]
    lib.fs2_op_layernorm_bwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fs2_op_layernorm_bwd_dropout.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_layernorm_bwd_masked.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_layernorm_head.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.fs2_op_layernorm_dropout.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_col_sum_ws_bytes.restype = sz
    lib.fs2_op_col_sum_ws_bytes.argtypes = [i32, i32, i32]
    lib.fs2_op_col_sum.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    lib.fs2_op_col_sum2.argtypes = [i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, vp]
    lib.fs2_op_col_sum_weighted.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.fs2_op_softmax_fwd.argtypes = [i32, vp, vp, vp, i32, i32, i32, f32, vp]
    lib.fs2_op_softmax_bwd.argtypes = [i32, vp, vp, vp, i32, i32, i32, f32, vp]
    lib.fs2_op_ew.argtypes = [i32, i32, vp, vp, vp, sz, f32, f32, vp]
    lib.fs2_op_scatter_rows_ws_bytes.restype = sz
    lib.fs2_op_scatter_rows_ws_bytes.argtypes = [i32, i32, i32]
    lib.fs2_op_scatter_rows.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_regulate_bwd.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_masked_loss_bwd.argtypes = [vp, vp, i32, vp, vp, vp, C.c_int64, i32, i32, f32, vp]
    lib.fs2_op_bucket_embed_utt.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp]
    lib.fs2_op_dropout.argtypes = [i32, vp, vp, sz, f32, C.c_uint64, C.c_uint64, vp]
    lib.fs2_op_row_dot.argtypes = [i32, vp, vp, vp, vp, vp, C.c_int64, i32, vp]
    lib.fs2_op_dwconv_dgrad.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_dwconv_wgrad_parts.restype = i32
    lib.fs2_op_dwconv_wgrad_parts.argtypes = [i32, i32]
    lib.fs2_op_dwconv_wgrad.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fs2_op_fold_conv2.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.fs2_op_unfold_conv2.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.fs2_op_transpose_weight.argtypes = [i32, vp, vp, i32, i32, i32, vp]
    lib.fs2_op_sum_sq_ws_bytes.restype = sz
    lib.fs2_op_sum_sq_ws_bytes.argtypes = [sz]
    lib.fs2_op_sum_sq.argtypes = [vp, sz, vp, vp, vp]
    lib.fs2_op_adamw.argtypes = [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, vp, f32, f32, vp]
    lib.fs2_op_adamw_shadow.argtypes = [vp, vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, vp, f32, f32, vp]
    lib.fs2_op_transpose_weight_tiles.restype = C.c_int64
    lib.fs2_op_transpose_weight_tiles.argtypes = [i32, i32, i32]
    lib.fs2_op_transpose_weight_batch.argtypes = [vp, i32, C.c_int64, vp]
    lib.fs2_op_bucket_embed_target.argtypes = [i32, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, i32, i32, i32, vp]
    if lib.fs2_abi_version() != FS2_ABI_VERSION:
        raise Fs2LibraryError("libfs2_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status: int, engine=None, what: str = ""):
    if status == FS2_OK:
        return
    lib = load()
    msg = lib.fs2_status_string(status).decode()
    if engine:
        detail = lib.fs2_last_error(engine).decode()
        if detail:
            msg = f"{msg}: {detail}"
    raise RuntimeError(f"fs2 {what} failed ({status}): {msg}")


def config_to_c(cfg, dtype: int) -> Fs2ConfigC:
    """lightningfastspeech2_amd.config.Fs2Config -> fs2_config."""
    from .config import DVECTOR_DIM, PE_MAX_LEN
    c = Fs2ConfigC()
    c.abi_version = FS2_ABI_VERSION
    c.dtype = dtype
    c.n_phones = cfg.n_phones
    c.hidden = cfg.hidden
    c.n_mels = cfg.n_mels
    c.dvec_dim = DVECTOR_DIM
    c.max_frames = cfg.max_frames
    c.pe_len = PE_MAX_LEN
    if cfg.encoder_layers > FS2_MAX_LAYERS or cfg.decoder_layers > FS2_MAX_LAYERS:
        raise ValueError("too many layers for the C ABI")
    if len(cfg.variances) > FS2_MAX_VARIANCES:
        raise ValueError("too many variances for the C ABI")
    c.enc_layers, c.enc_heads = cfg.encoder_layers, cfg.encoder_head
    c.enc_filter, c.enc_depthwise = cfg.encoder_conv_filter_size, int(cfg.encoder_depthwise_conv)
    for i in range(cfg.encoder_layers):
        c.enc_kernels[i] = cfg.encoder_kernel_sizes[i]
    c.dec_layers, c.dec_heads = cfg.decoder_layers, cfg.decoder_head
    c.dec_filter, c.dec_depthwise = cfg.decoder_conv_filter_size, int(cfg.decoder_depthwise_conv)
    for i in range(cfg.decoder_layers):
        c.dec_kernels[i] = cfg.decoder_kernel_sizes[i]
    c.n_variances = len(cfg.variances)
    for i, v in enumerate(cfg.variances):
        name = v.encode()
        if len(name) >= FS2_NAME_LEN:
            raise ValueError("variance name too long")
        c.var_names[i].value = name
        c.var_nlayers[i] = cfg.variance_nlayers[i]
        c.var_kernel[i] = cfg.variance_kernel_size[i]
        cwt = cfg.is_cwt(i)  # the CWT head bucketises its recomposed log-domain signal directly (model.py:427-428)
        c.var_cwt[i] = int(cwt)
        c.var_level[i] = int(cfg.is_phone_level(i))
        c.var_mean[i] = 0.0 if cwt else cfg.stats[v]["mean"]
        c.var_std[i] = 1.0 if cwt else cfg.stats[v]["std"]
    c.var_filter, c.var_nbins = cfg.variance_filter_size, cfg.variance_nbins
    c.var_depthwise = int(cfg.variance_depthwise_conv)
    if len(cfg.priors) > FS2_MAX_PRIORS:
        raise ValueError("too many priors for the C ABI")
    c.n_priors = len(cfg.priors)
    for i, pr in enumerate(cfg.priors):
        c.prior_names[i].value = pr.encode()
    c.dur_nlayers, c.dur_kernel = cfg.duration_nlayers, cfg.duration_kernel_size
    c.dur_filter, c.dur_depthwise = cfg.duration_filter_size, int(cfg.duration_depthwise_conv)
    return c
