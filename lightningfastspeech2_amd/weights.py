"""state_dict naming of the reference model and deterministic synthetic weights.

``state_dict_spec`` lists every tensor ``FastSpeech2.state_dict()`` of the reference holds for the
forward path with the reference's own key names (modules built at
/root/reference/litfass/fastspeech2/fastspeech2.py:242-438 from model.py).  The HIP engine's
``fs2_load_weight`` is keyed by exactly these names, so a Lightning checkpoint's ``state_dict``
can be fed through unchanged.

``synth_state_dict`` draws weights of those shapes from the distributions PyTorch's default
initialisers use, with ``numpy.random.RandomState`` (a frozen bit-stream) in spec order, so the
same (config, seed) gives the same weights in this container, on the GPU box, and inside
tools/gen_golden.py where they are loaded into the *reference* modules.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from .config import DVECTOR_DIM, PE_MAX_LEN, Fs2Config


def _conformer_layer_spec(prefix, H, heads, F, k, depthwise, out):
    out[f"{prefix}.self_attn.in_proj_weight"] = (3 * H, H)
    out[f"{prefix}.self_attn.in_proj_bias"] = (3 * H,)
    out[f"{prefix}.self_attn.out_proj.weight"] = (H, H)
    out[f"{prefix}.self_attn.out_proj.bias"] = (H,)
    out[f"{prefix}.norm1.weight"] = (H,)
    out[f"{prefix}.norm1.bias"] = (H,)
    out[f"{prefix}.norm2.weight"] = (H,)
    out[f"{prefix}.norm2.bias"] = (H,)
    if depthwise:  # model.py:73-93
        out[f"{prefix}.conv1.0.weight"] = (H, 1, k)
        out[f"{prefix}.conv1.0.bias"] = (H,)
        out[f"{prefix}.conv1.1.weight"] = (F, H, 1)
        out[f"{prefix}.conv1.1.bias"] = (F,)
        out[f"{prefix}.conv2.0.weight"] = (F, F // H, 1)  # groups=conv_in over F channels
        out[f"{prefix}.conv2.0.bias"] = (F,)
        out[f"{prefix}.conv2.1.weight"] = (H, F, 1)
        out[f"{prefix}.conv2.1.bias"] = (H,)
    else:  # model.py:94-106 ; conv2 kernel is fixed to 1 (fastspeech2.py:282,370)
        out[f"{prefix}.conv1.weight"] = (F, H, k)
        out[f"{prefix}.conv1.bias"] = (F,)
        out[f"{prefix}.conv2.weight"] = (H, F, 1)
        out[f"{prefix}.conv2.bias"] = (H,)


CWT_SCALES = 10  # VariancePredictor(cwt=True).linear = Linear(filter, 10) (model.py:505-508), CWT(n_scales=10) (dataset/cwt.py:25)


def _predictor_spec(prefix, nlayers, cin, filt, k, depthwise, out, n_out=1):
    # VariancePredictor / VarianceConvolutionLayer, model.py:482-561
    for j in range(nlayers):
        p = f"{prefix}.layers.{j}.layers"
        if depthwise:
            out[f"{p}.0.module.0.weight"] = (cin, 1, k)
            out[f"{p}.0.module.0.bias"] = (cin,)
            out[f"{p}.0.module.1.weight"] = (filt, cin, 1)
            out[f"{p}.0.module.1.bias"] = (filt,)
        else:
            out[f"{p}.0.module.weight"] = (filt, cin, k)
            out[f"{p}.0.module.bias"] = (filt,)
        out[f"{p}.2.weight"] = (filt,)
        out[f"{p}.2.bias"] = (filt,)
    out[f"{prefix}.linear.weight"] = (n_out, filt)
    out[f"{prefix}.linear.bias"] = (n_out,)


def state_dict_spec(cfg: Fs2Config) -> "OrderedDict[str, tuple]":
    H = cfg.hidden
    out: "OrderedDict[str, tuple]" = OrderedDict()
    out["phone_embedding.weight"] = (cfg.n_phones, H)
    for i in range(cfg.encoder_layers):
        _conformer_layer_spec(f"encoder.layers.{i}", H, cfg.encoder_head, cfg.encoder_conv_filter_size,
                              cfg.encoder_kernel_sizes[i], cfg.encoder_depthwise_conv, out)
    out["positional_encoding.pe"] = (1, PE_MAX_LEN, H)
    _predictor_spec("variance_adaptor.duration_predictor", cfg.duration_nlayers, H,
                    cfg.duration_filter_size, cfg.duration_kernel_size, cfg.duration_depthwise_conv, out)
    for vi, var in enumerate(cfg.variances):
        p = f"variance_adaptor.encoders.{var}"
        out[f"{p}.bins"] = (cfg.variance_nbins - 1,)
        out[f"{p}.embedding.weight"] = (cfg.variance_nbins, H)
        _predictor_spec(f"{p}.predictor", cfg.variance_nlayers[vi], H, cfg.variance_filter_size,
                        cfg.variance_kernel_size[vi], cfg.variance_depthwise_conv, out,
                        n_out=CWT_SCALES if cfg.is_cwt(vi) else 1)
        if cfg.is_cwt(vi):  # mean_std_linear = nn.Linear(filter_size, 2) (model.py:402-404)
            out[f"{p}.mean_std_linear.weight"] = (2, cfg.variance_filter_size)
            out[f"{p}.mean_std_linear.bias"] = (2,)
    for i in range(cfg.decoder_layers):
        _conformer_layer_spec(f"decoder.layers.{i}", H, cfg.decoder_head, cfg.decoder_conv_filter_size,
                              cfg.decoder_kernel_sizes[i], cfg.decoder_depthwise_conv, out)
    out["linear.weight"] = (cfg.n_mels, H)
    out["linear.bias"] = (cfg.n_mels,)
    out["speaker_embedding.projection.weight"] = (H, DVECTOR_DIM)
    out["speaker_embedding.projection.bias"] = (H,)
    for pr in cfg.priors:  # PriorEmbedding, model.py:146-164 (ModuleDict at fastspeech2.py:416-424)
        out[f"prior_embeddings.{pr}.bins"] = (cfg.variance_nbins - 1,)
        out[f"prior_embeddings.{pr}.embedding.weight"] = (cfg.variance_nbins, H)
    return out


def positional_table(H: int, max_len: int = PE_MAX_LEN) -> np.ndarray:
    """PositionalEncoding buffer, model.py:43-50: pe[p, 2i] = sin(p * exp(-2i ln(1e4)/H)),
    pe[p, 2i+1] = cos(.).  Evaluated in float64 and rounded once to float32 so the table is the
    same bits on every host (torch's float32 sin/cos differ by an ulp between CPU dispatch paths);
    it travels in the state_dict (``positional_encoding.pe`` is a registered buffer), so the
    reference, the oracle and the engine all consume this exact tensor."""
    position = np.arange(max_len, dtype=np.float64)[:, None]
    div_term = np.exp(np.arange(0, H, 2, dtype=np.float64) * (-math.log(10000.0) / H))
    pe = np.zeros((max_len, H), np.float64)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe[None].astype(np.float32)


def variance_bins(cfg: Fs2Config, var: str) -> np.ndarray:
    """VarianceEncoder.bins / PriorEmbedding.bins = linspace(min, max, nbins-1), model.py:397-400,
    150-153 (float64 arithmetic, one rounding; carried in the state_dict as the ``bins`` parameter).
    ``var`` is a variance name or ``"<prior>_prior"``."""
    st = cfg.stats[var]
    n = cfg.variance_nbins - 1
    lo, hi = float(st["min"]), float(st["max"])
    if var in cfg.variances and cfg.is_cwt(cfg.variances.index(var)):  # min = np.log(min), max = np.log(max) (model.py:394-396)
        lo, hi = float(np.log(lo)), float(np.log(hi))
    if n == 1:
        return np.array([lo], np.float32)
    return (lo + (hi - lo) * (np.arange(n, dtype=np.float64) / (n - 1))).astype(np.float32)


def synth_state_dict(cfg: Fs2Config, seed: int = 0, *, randomize_norm: bool = False,
                     duration_bias: float | None = None, duration_weight_scale: float = 1.0
                     ) -> "OrderedDict[str, np.ndarray]":
    """Random-init weights of the reference architecture (float32 numpy, reference key names).

    Distributions follow torch's defaults: Linear/Conv1d U(+-1/sqrt(fan_in)) for weight and bias
    (kaiming_uniform(a=sqrt 5)), MHA in_proj xavier-uniform with zero biases, Embedding N(0,1)
    with the padding row zeroed, LayerNorm 1/0 (``randomize_norm`` perturbs them so tests can see
    gamma/beta mix-ups).  ``duration_bias`` overrides the duration head bias (weight scaled by
    ``duration_weight_scale``): bench uses weight 0 / bias ln 7 -> 6 frames per phone.
    """
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    fan_in = 1
    for name, shape in state_dict_spec(cfg).items():
        if name == "positional_encoding.pe":
            sd[name] = positional_table(cfg.hidden)
            continue
        if name.endswith(".bins"):
            parts = name.split(".")
            sd[name] = variance_bins(cfg, parts[2] if parts[0] == "variance_adaptor" else f"{parts[1]}_prior")
            continue
        if name.endswith("embedding.weight"):
            w = rs.standard_normal(shape).astype(np.float32)
            if name == "phone_embedding.weight":
                w[0] = 0.0  # padding_idx=0
            sd[name] = w
            continue
        leaf = name.rsplit(".", 1)[1]
        is_norm = ".norm1." in name or ".norm2." in name or name.rsplit(".", 2)[1] == "2"
        if is_norm:
            if leaf == "weight":
                w = np.ones(shape, np.float32)
                if randomize_norm:
                    w += 0.2 * rs.standard_normal(shape).astype(np.float32)
            else:
                w = np.zeros(shape, np.float32)
                if randomize_norm:
                    w += 0.1 * rs.standard_normal(shape).astype(np.float32)
            sd[name] = w
            continue
        if name.endswith("in_proj_weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))  # xavier_uniform_
            sd[name] = rs.uniform(-bound, bound, shape).astype(np.float32)
            continue
        if name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            w = np.zeros(shape, np.float32)  # nn.MultiheadAttention._reset_parameters
            if randomize_norm:
                w += 0.05 * rs.standard_normal(shape).astype(np.float32)
            sd[name] = w
            continue
        # Linear / Conv1d weight or bias: fan_in from the matching weight shape
        if leaf == "weight":
            fan_in = int(np.prod(shape[1:]))  # remembered for the bias that follows
        bound = 1.0 / math.sqrt(fan_in)
        sd[name] = rs.uniform(-bound, bound, shape).astype(np.float32)
    if duration_bias is not None:
        p = "variance_adaptor.duration_predictor.linear"
        sd[f"{p}.weight"] = (sd[f"{p}.weight"] * np.float32(duration_weight_scale)).astype(np.float32)
        sd[f"{p}.bias"] = np.full((1,), duration_bias, np.float32)
    return sd


def synth_inputs(cfg: Fs2Config, B: int, L: int, seed: int = 1234, *, lengths=None):
    """Synthetic batch in the reference's collate format (datasets.py:852-882): ``phones`` int64
    (B, L) zero-padded on the right, ``speaker`` float32 (B, 256) unit-norm d-vectors, and one
    ``priors_<p>`` float32 (B,) per configured prior, drawn inside its [min, max] range."""
    rs = np.random.RandomState(seed)
    phones = rs.randint(1, cfg.n_phones, size=(B, L)).astype(np.int64)
    if lengths is not None:
        for b, n in enumerate(lengths):
            phones[b, int(n):] = 0
    spk = np.random.RandomState(seed + 1).standard_normal((B, DVECTOR_DIM)).astype(np.float32)
    spk /= np.linalg.norm(spk, axis=1, keepdims=True)
    out = {"phones": phones, "speaker": spk.astype(np.float32)}
    prs = np.random.RandomState(seed + 2)
    for pr in cfg.priors:
        st = cfg.stats[f"{pr}_prior"]
        out[f"priors_{pr}"] = prs.uniform(st["min"] - 0.1, st["max"] + 0.1, size=B).astype(np.float32)
    return out
