"""CPU ORACLE — test infrastructure, NOT product code.

A from-scratch functional restatement of the reference's mel forward,
``FastSpeech2.forward(targets, inference=True)``
(/root/reference/litfass/fastspeech2/fastspeech2.py:636-731) and the blocks it is built from
(/root/reference/litfass/fastspeech2/model.py), written against plain CPU tensor ops.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file; the product path (``lightningfastspeech2_amd``) never does and fails loudly when the HIP
library is missing.

Pinning: the reference has no tests or golden vectors of its own, so this oracle is
pinned against outputs of the reference itself: ``tools/gen_golden.py`` imports the real
``litfass`` modules in the build container, runs the unmodified ``FastSpeech2.forward`` and commits
inputs/outputs/intermediates under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
file against every one of them (and, where /root/reference is present, against the live import on
randomised hparams).

The arithmetic lives in PyTorch ops (the reference pins pytorch 1.10, environment.yaml:104; here
2.10 — same op semantics).  torch-1.10 container semantics are restated explicitly: the
``nn.TransformerEncoder`` loop is ``for layer: x = layer(x, src_key_padding_mask=mask)`` with no
final norm, each layer post-LN (model.py:113-115).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, name) -> torch.Tensor:
    v = sd[name]
    if isinstance(v, torch.Tensor) and v.requires_grad:  # oracle/train_cpu.py: leaves of the autograd restatement of the training step
        return v
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.detach().to(torch.float32) if v.is_floating_point() else v


def positional_encoding(x: torch.Tensor, pe: torch.Tensor) -> torch.Tensor:
    """model.py:53-55 — x + pe[:, :len]; dropout is identity in eval."""
    return x + pe[:, : x.size(1), :]


def speaker_embedding(sd, dvec: torch.Tensor) -> torch.Tensor:
    """model.py:137-143 — relu(Linear(256->H)(dvec)), broadcast over time by the caller."""
    w = _t(sd, "speaker_embedding.projection.weight")
    b = _t(sd, "speaker_embedding.projection.bias")
    return torch.relu(F.linear(dvec, w, b))  # (B, H)


def self_attention(x, w_in, b_in, w_out, b_out, heads: int, key_padding_mask) -> torch.Tensor:
    """nn.MultiheadAttention forward as nn.TransformerEncoderLayer._sa_block calls it
    (model.py:114): packed in-proj [Wq;Wk;Wv], q scaled by 1/sqrt(d) before QK^T,
    padded KEYS get -inf, padded queries are still computed."""
    B, S, H = x.shape
    d = H // heads
    qkv = F.linear(x, w_in, b_in)
    q, k, v = qkv.split(H, dim=-1)
    q = q.view(B, S, heads, d).transpose(1, 2) * (1.0 / math.sqrt(d))
    k = k.view(B, S, heads, d).transpose(1, 2)
    v = v.view(B, S, heads, d).transpose(1, 2)
    scores = q @ k.transpose(-1, -2)  # (B, h, S, S)
    scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, S, H)
    return F.linear(o, w_out, b_out)


def conv_ffn(sd, prefix: str, x, H: int, depthwise: bool) -> torch.Tensor:
    """ConformerEncoderLayer._ff_block, model.py:118-122 with the two conv variants of
    model.py:73-106: conv2(relu(conv1(x^T)))^T, padding='same' (zero), unmasked."""
    y = x.transpose(1, 2)
    if depthwise:
        w = _t(sd, f"{prefix}.conv1.0.weight")
        y = F.conv1d(y, w, _t(sd, f"{prefix}.conv1.0.bias"), padding="same", groups=H)
        y = F.conv1d(y, _t(sd, f"{prefix}.conv1.1.weight"), _t(sd, f"{prefix}.conv1.1.bias"))
        y = torch.relu(y)
        # grouped k=1 conv with groups=conv_in over F channels (model.py:84-91)
        y = F.conv1d(y, _t(sd, f"{prefix}.conv2.0.weight"), _t(sd, f"{prefix}.conv2.0.bias"),
                     padding="same", groups=H)
        y = F.conv1d(y, _t(sd, f"{prefix}.conv2.1.weight"), _t(sd, f"{prefix}.conv2.1.bias"))
    else:
        y = F.conv1d(y, _t(sd, f"{prefix}.conv1.weight"), _t(sd, f"{prefix}.conv1.bias"), padding="same")
        y = torch.relu(y)
        y = F.conv1d(y, _t(sd, f"{prefix}.conv2.weight"), _t(sd, f"{prefix}.conv2.bias"), padding="same")
    return y.transpose(1, 2)


def conformer_layer(sd, prefix: str, x, heads: int, depthwise: bool, key_padding_mask) -> torch.Tensor:
    """ConformerEncoderLayer.forward, norm_first=False branch (model.py:113-115), LN eps 1e-5."""
    H = x.shape[-1]
    a = self_attention(x, _t(sd, f"{prefix}.self_attn.in_proj_weight"), _t(sd, f"{prefix}.self_attn.in_proj_bias"),
                       _t(sd, f"{prefix}.self_attn.out_proj.weight"), _t(sd, f"{prefix}.self_attn.out_proj.bias"),
                       heads, key_padding_mask)
    x = F.layer_norm(x + a, (H,), _t(sd, f"{prefix}.norm1.weight"), _t(sd, f"{prefix}.norm1.bias"), 1e-5)
    f = conv_ffn(sd, prefix, x, H, depthwise)
    x = F.layer_norm(x + f, (H,), _t(sd, f"{prefix}.norm2.weight"), _t(sd, f"{prefix}.norm2.bias"), 1e-5)
    return x


def variance_predictor(sd, prefix: str, x, nlayers: int, kernel: int, depthwise: bool, mask, cwt: bool = False):
    """VariancePredictor.forward (model.py:510-522) over VarianceConvolutionLayer (model.py:524-561):
    n x [conv(k, pad (k-1)//2) (dense, or dw + pw 1x1) -> ReLU -> LayerNorm] -> Linear(.,1) ->
    squeeze -> masked_fill(mask, 0)."""
    y = x
    for j in range(nlayers):
        p = f"{prefix}.layers.{j}.layers"
        z = y.transpose(1, 2)
        pad = (kernel - 1) // 2
        if depthwise:
            w0 = _t(sd, f"{p}.0.module.0.weight")
            z = F.conv1d(z, w0, _t(sd, f"{p}.0.module.0.bias"), padding=pad, groups=w0.shape[0])
            z = F.conv1d(z, _t(sd, f"{p}.0.module.1.weight"), _t(sd, f"{p}.0.module.1.bias"))
        else:
            z = F.conv1d(z, _t(sd, f"{p}.0.module.weight"), _t(sd, f"{p}.0.module.bias"), padding=pad)
        z = torch.relu(z.transpose(1, 2))
        g = _t(sd, f"{p}.2.weight")
        y = F.layer_norm(z, (g.shape[0],), g, _t(sd, f"{p}.2.bias"), 1e-5)
    out = F.linear(y, _t(sd, f"{prefix}.linear.weight"), _t(sd, f"{prefix}.linear.bias"))
    if cwt:  # Linear(filter, 10), mask stacked over the 10 scales, out_conv handed back (model.py:505-522)
        return out.masked_fill(mask[..., None], 0), y
    return out.squeeze(-1).masked_fill(mask, 0)


def round_durations(duration_pred: torch.Tensor, src_mask: torch.Tensor):
    """model.py:299-309 — round-half-even(exp(p)-1), clamp >=0, .int(); zero-duration guard:
    if sum over valid phones <= n_valid // 2, every valid phone gets duration 1."""
    d = torch.clamp(torch.round(torch.exp(duration_pred) - 1), min=0).int()
    guarded = []
    for i in range(len(d)):
        valid = ~src_mask[i]
        if d[i][valid].sum() <= valid.sum() // 2:
            d[i][valid] = 1
            guarded.append(i)
    return d, guarded


def length_regulator(x: torch.Tensor, durations: torch.Tensor, max_length: float):
    """LengthRegulator.forward (model.py:349-370): per-utterance repeat_interleave,
    T = min(max total, int(max_length)), zero pad / truncate, mask = t >= total (untruncated)."""
    B, L, H = x.shape
    totals = durations.long().sum(dim=1)
    T = int(min(int(totals.max()), int(max_length)))
    out = x.new_zeros(B, T, H)
    for b in range(B):
        rep = torch.repeat_interleave(x[b], durations[b].long(), dim=0)[:T]
        out[b, : rep.shape[0]] = rep
    mask = ~(torch.arange(T)[None, :] < totals[:, None])
    return out, mask


def variance_encoder(sd, cfg, var_index: int, x, mask, tgt=None):
    """VarianceEncoder.forward, non-CWT branch (model.py:409-441): with ``tgt`` (teacher forcing,
    model.py:417-422) the embedding comes from bucketize(tgt*std+mean), the prediction is still
    computed and returned."""
    var = cfg.variances[var_index]
    p = f"variance_adaptor.encoders.{var}"
    if cfg.variance_transforms[var_index] == "cwt":
        return _variance_encoder_cwt(sd, cfg, var_index, x, mask, tgt)
    pred = variance_predictor(sd, f"{p}.predictor", x, cfg.variance_nlayers[var_index],
                              cfg.variance_kernel_size[var_index], cfg.variance_depthwise_conv, mask)
    st = cfg.stats[var]
    bucket_value = (pred if tgt is None else tgt) * st["std"] + st["mean"]  # model.py:434 / :421
    idx = torch.bucketize(bucket_value, _t(sd, f"{p}.bins"))  # right=False
    emb = F.embedding(idx, _t(sd, f"{p}.embedding.weight"))
    return pred, emb, idx


def _variance_encoder_cwt(sd, cfg, var_index: int, x, mask, tgt=None):
    """VarianceEncoder.forward, CWT branch (model.py:412-431,445-461) with CWT.recompose (dataset/cwt.py:18-21,49-50):
    a 10-scale wavelet spectrogram per frame, utterance-level (mean, std) from the time-mean of the last conv layer,
    signal = z-normalised (unbiased std, +1e-7) sum over the scales, * std + mean  -> log-domain pitch, bucketised against
    log-spaced bins.  Returns (dict as the reference does, embedding, bucket indices); with ``tgt`` (the raw signal,
    targets["variances_<var>_signal"]) the embedding comes from bucketize(log(tgt))."""
    var = cfg.variances[var_index]
    p = f"variance_adaptor.encoders.{var}"
    spec, out_conv = variance_predictor(sd, f"{p}.predictor", x, cfg.variance_nlayers[var_index],
                                        cfg.variance_kernel_size[var_index], cfg.variance_depthwise_conv, mask, cwt=True)
    ms = F.linear(out_conv.mean(dim=1), _t(sd, f"{p}.mean_std_linear.weight"), _t(sd, f"{p}.mean_std_linear.bias"))
    mean, std = ms[:, 0], ms[:, 1]
    bins = _t(sd, f"{p}.bins")
    if tgt is not None:
        idx = torch.bucketize(torch.log(tgt), bins)
        return {"spectrogram": spec, "mean": mean, "std": std}, F.embedding(idx, _t(sd, f"{p}.embedding.weight")), idx
    sig = spec.sum(dim=-1)                                            # wavelet_recomposition: sum over the scales
    sig = (sig - sig.mean(dim=1, keepdim=True)) / (sig.std(dim=1, keepdim=True) + 1e-7)  # torch .std(): unbiased
    pred = sig * std[:, None] + mean[:, None]
    idx = torch.bucketize(pred, bins)
    res = {"reconstructed_signal": torch.exp(pred), "spectrogram": spec, "mean": mean, "std": std}
    return res, F.embedding(idx, _t(sd, f"{p}.embedding.weight")), idx


def prior_embedding(sd, cfg, prior: str, values: torch.Tensor) -> torch.Tensor:
    """PriorEmbedding.forward, model.py:160-164 — relu(Emb[bucketize(prior, bins)]), one row per
    utterance, broadcast over time by the caller."""
    p = f"prior_embeddings.{prior}"
    idx = torch.bucketize(values, _t(sd, f"{p}.bins"))
    return torch.relu(F.embedding(idx, _t(sd, f"{p}.embedding.weight")))


def forward(sd, cfg, phones, speaker, *, return_intermediates: bool = False,
            force_durations: Optional[torch.Tensor] = None, priors: Optional[dict] = None,
            teacher_targets: Optional[dict] = None, frames_hook=None) -> Dict[str, torch.Tensor]:
    """FastSpeech2.forward(targets, inference=True), fastspeech2.py:636-731 (mel path only; the
    fastdiff_var branch :733-736 is broken at HEAD and not part of mel — SURVEY §0.6)."""
    phones = torch.as_tensor(phones).long()
    speaker = torch.as_tensor(speaker).float()
    H = cfg.hidden
    pe = _t(sd, "positional_encoding.pe")
    src_mask = phones.eq(0)                                                   # :651
    x = F.embedding(phones, _t(sd, "phone_embedding.weight"), padding_idx=0)  # :653
    x = positional_encoding(x, pe)                                            # :655
    spk = speaker_embedding(sd, speaker)                                      # :657-660
    x = x + spk[:, None, :]
    inter = {}
    for i in range(cfg.encoder_layers):                                       # :685 (torch-1.10 loop)
        x = conformer_layer(sd, f"encoder.layers.{i}", x, cfg.encoder_head,
                            cfg.encoder_depthwise_conv, src_mask)
    inter["encoder_out"] = x
    for pr in cfg.priors:                                                     # :687-692
        v = torch.as_tensor(np.asarray(priors[f"priors_{pr}"])).float()
        x = x + prior_embedding(sd, cfg, pr, v)[:, None, :]
    # ---- VarianceAdaptor.forward, model.py:249-341 ----
    dur_pred = variance_predictor(sd, "variance_adaptor.duration_predictor", x, cfg.duration_nlayers,
                                  cfg.duration_kernel_size, cfg.duration_depthwise_conv, src_mask)
    result = {}

    def teacher_target(vi, var):  # model.py:278-286 / :317-325 (a CWT variance is forced with its raw signal)
        if teacher_targets is None:
            return None
        key = f"variances_{var}_signal" if cfg.variance_transforms[vi] == "cwt" else f"variances_{var}"
        return torch.as_tensor(np.asarray(teacher_targets[key])).float()
    for vi, var in enumerate(cfg.variances):                                  # phone-level variances, model.py:276-294:
        if cfg.variance_levels[vi] != "phone":                               # after the duration predictor, before the regulator
            continue
        pred, emb, idx = variance_encoder(sd, cfg, vi, x, src_mask, teacher_target(vi, var))
        result[f"variances_{var}"] = pred
        inter[f"bucket_{var}"] = idx
        x = x + emb
    if teacher_targets is not None:   # inference=False: targets["duration"], no rounding/guard (model.py:296-297)
        dur_rounded, guarded = torch.as_tensor(np.asarray(teacher_targets["duration"])), []
    elif force_durations is None:
        dur_rounded, guarded = round_durations(dur_pred, src_mask)
    else:
        dur_rounded, guarded = torch.as_tensor(force_durations).int(), []
    x, tgt_mask = length_regulator(x, dur_rounded, cfg.max_length * cfg.sampling_rate / cfg.hop_length)
    if frames_hook is not None:  # data-parallel global-pad mode: pad to the frame count of the whole batch
        Tg = int(frames_hook(x.shape[1]))  # (what pad_sequence over ALL utterances would have produced)
        if Tg > x.shape[1]:
            x = F.pad(x, (0, 0, 0, Tg - x.shape[1]))
            tgt_mask = F.pad(tgt_mask, (0, Tg - tgt_mask.shape[1]), value=True)
    inter["regulated"] = x
    for vi, var in enumerate(cfg.variances):                                  # frame-level variances, model.py:315-333
        if cfg.variance_levels[vi] != "frame":
            continue
        pred, emb, idx = variance_encoder(sd, cfg, vi, x, tgt_mask, teacher_target(vi, var))
        result[f"variances_{var}"] = pred
        inter[f"bucket_{var}"] = idx
        x = x + emb
    inter["adaptor_out"] = x
    y = positional_encoding(x, pe)                                            # :705
    y = y + spk[:, None, :]                                                   # :707-718
    for i in range(cfg.decoder_layers):                                       # :719-721
        y = conformer_layer(sd, f"decoder.layers.{i}", y, cfg.decoder_head,
                            cfg.decoder_depthwise_conv, tgt_mask)
    inter["decoder_out"] = y
    mel = F.linear(y, _t(sd, "linear.weight"), _t(sd, "linear.bias"))         # :723
    result.update({
        "mel": mel,
        "duration_prediction": dur_pred,
        "duration_rounded": dur_rounded,
        "src_mask": src_mask,
        "tgt_mask": tgt_mask,
    })
    if return_intermediates:
        result["_intermediates"] = inter
        result["_zero_duration_guard"] = guarded
    return result
