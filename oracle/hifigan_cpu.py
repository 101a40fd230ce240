"""CPU ORACLE for the HiFi-GAN generator — test infrastructure, NOT product code.

A from-scratch functional restatement of ``Generator.forward``
(/root/reference/litfass/third_party/hifigan/models.py:145-162) with ResBlock "1" (models.py:98-104)
on plain (weight-norm-removed) weights, as ``Synthesiser.__call__`` drives it
(/root/reference/litfass/third_party/hifigan/__init__.py:37-43): one utterance at a time,
mel ``(T, 80)`` -> samples ``(T*256,)`` in [-1, 1].  Only ``tests/``, ``__graft_entry__.smoke()`` and
benchmark ``cpu_baseline`` legs may import this file.

Pinning: the reference ships no tests, golden vectors or generator weights (the ``generator_*.pth.tar``
blobs are absent), so this oracle is pinned against outputs of the reference's own ``Generator`` class
run in the build container on seeded random weights: ``tools/gen_golden_hifigan.py`` ->
``tests/golden/hifigan_*.npz``, checked by ``tests/test_hifigan_oracle.py``.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # models.py:7


def _t(sd, name) -> torch.Tensor:
    v = sd[name]
    return torch.from_numpy(np.asarray(v)).float() if not isinstance(v, torch.Tensor) else v.detach().float()


def resblock(sd, prefix: str, x: torch.Tensor, kernel: int, dilations) -> torch.Tensor:
    """ResBlock.forward (models.py:98-104): x = c2(lrelu(c1(lrelu(x)))) + x for the three pairs."""
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _t(sd, f"{prefix}.convs1.{m}.weight"), _t(sd, f"{prefix}.convs1.{m}.bias"),
                      dilation=d, padding=(kernel * d - d) // 2)          # get_padding, models.py:16-17
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, _t(sd, f"{prefix}.convs2.{m}.weight"), _t(sd, f"{prefix}.convs2.{m}.bias"),
                      dilation=1, padding=(kernel - 1) // 2)
        x = xt + x
    return x


def generator(sd: Dict, cfg, mel_ct: torch.Tensor, return_stages: bool = False):
    """mel_ct: (1, n_mels, T) -> (1, 1, T*hop).  models.py:145-162."""
    nk = len(cfg.resblock_kernel_sizes)
    stages = []
    x = F.conv1d(mel_ct, _t(sd, "conv_pre.weight"), _t(sd, "conv_pre.bias"), padding=3)
    stages.append(x)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _t(sd, f"ups.{i}.weight"), _t(sd, f"ups.{i}.bias"), stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            r = resblock(sd, f"resblocks.{i * nk + j}", x, rk, rd)
            xs = r if xs is None else xs + r
        x = xs / nk
        stages.append(x)
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:158)
    x = F.conv1d(x, _t(sd, "conv_post.weight"), _t(sd, "conv_post.bias"), padding=3)
    x = torch.tanh(x)
    return (x, stages) if return_stages else x


def synthesize(sd: Dict, cfg, mel: torch.Tensor, lengths: Optional[torch.Tensor] = None, return_stages: bool = False):
    """Batched convenience over the reference's per-utterance call: mel (B, T, n_mels), optional valid
    frame counts; every utterance runs alone on its first lengths[b] frames (generator.py:163-170).
    Returns wav (B, T*hop) with zeros past each utterance (and per-utterance stage outputs, time-major)."""
    mel = torch.as_tensor(mel, dtype=torch.float32)
    B, T, _ = mel.shape
    hop = int(np.prod(cfg.upsample_rates))
    wav = torch.zeros(B, T * hop)
    all_stages = []
    with torch.no_grad():
        for b in range(B):
            n = T if lengths is None else int(lengths[b])
            if n == 0:
                all_stages.append([])
                continue
            out = generator(sd, cfg, mel[b, :n].T.unsqueeze(0), return_stages)
            y, st = out if return_stages else (out, [])
            wav[b, :n * hop] = y[0, 0]
            all_stages.append([s[0].T.contiguous() for s in st])  # (T_i, C_i)
    return (wav, all_stages) if return_stages else wav
