#!/usr/bin/env python
"""Benchmark of the HiFi-GAN generator on the mel forward's own output shape:
B utterances x T frames of synthetic mel -> B x T*256 samples, inputs resident in HBM.

    python tools/bench_vocoder.py [--batch 32 --frames 1536 --steps 5 --warmup 2 --precision bf16]

Prints ONE JSON line: samples/s and mel-frames/s of the whole batch, the aggregate MFMA roofline
(algorithmic FLOPs of every conv in Generator.forward / wall time per pass) and a CPU baseline (the
oracle on a bounded sample).  Random-init weights (the reference's generator checkpoints are absent).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from lightningfastspeech2_amd.hifigan import HifiGan, HifiGanConfig, synth_state_dict

MFMA_PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}


def flops_per_frame(cfg):
    """2*MAC of every conv per mel frame (models.py:145-162), transposed convs at their true k/stride taps."""
    ch, fl, up = cfg.channels(), 0.0, 1
    fl += 2.0 * cfg.num_mels * ch[0] * 7
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        up *= u
        fl += up * 2.0 * ch[i] * ch[i + 1] * (k / u)
        for rk in cfg.resblock_kernel_sizes:
            fl += up * 6 * 2.0 * ch[i + 1] ** 2 * rk
    fl += up * 2.0 * ch[-1] * 7
    return fl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1536)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=192)
    ap.add_argument("--no-fused-resblock", action="store_true", help="A/B: narrow-stage resblocks conv by conv")
    ap.add_argument("--fused-mode", type=int, default=1, help="A/B: 1 auto, 2 pairs only, 4 tallest tiles only")
    ap.add_argument("--lds-limit", type=int, default=0, help="tuning: KiB cap on a conv workgroup's LDS slab (0 = heuristic)")
    a = ap.parse_args()
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 0)
    g = HifiGan(cfg, sd, precision=a.precision)
    from lightningfastspeech2_amd import _lib
    if a.lds_limit:
        _lib.load().fs2_op_set_vocoder_lds_limit(a.lds_limit)
    _lib.load().fs2_op_set_vocoder_fused_resblock(0 if a.no_fused_resblock else a.fused_mode)
    rs = np.random.RandomState(1234)
    mel = torch.from_numpy((rs.standard_normal((a.batch, a.frames, 80)) * 1.5 - 4.0).astype(np.float32)).cuda()
    for _ in range(a.warmup):
        wav = g.synthesize(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wav = g.synthesize(mel)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.steps
    assert bool(torch.isfinite(wav).all())
    frames = a.batch * a.frames
    fl = flops_per_frame(cfg) * frames
    line = {"metric": "audio samples/sec, HiFi-GAN V1 generator", "value": frames * cfg.hop / el, "unit": "samples/s",
            "mel_frames_per_s": frames / el, "ms_per_step": el * 1e3, "steps": a.steps, "warmup": a.warmup, "n_gpus": 1,
            "rtf": el / (frames * cfg.hop / cfg.sampling_rate), "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"HiFi-GAN V1 (config.json: rates 8,8,2,2, 512 ch, resblock 1, k 3/7/11), batch {a.batch} x "
                                   f"{a.frames} mel frames -> {a.frames * cfg.hop} samples each, random-init weights",
                       "params": int(sum(np.prod(v.shape) for v in sd.values()))},
            "roofline": {"bound": "mfma", "achieved": fl / el / 1e12, "peak": MFMA_PEAK[a.precision] / 1e12, "unit": "TFLOP/s",
                         "frac": fl / el / MFMA_PEAK[a.precision], "traffic": None,
                         "kernel": "vocoder_conv_kernel, all 78 launches of one pass (aggregate)",
                         "flops_per_pass": fl, "mflop_per_frame": flops_per_frame(cfg) / 1e6}}
    if not a.no_cpu_baseline:
        from oracle import hifigan_cpu  # cpu_baseline leg only
        m = mel[:1, :a.cpu_frames].cpu()
        best, bt = 0.0, 0
        for nt in (8, 16, 32, 64):
            if nt > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(nt)
            hifigan_cpu.synthesize(sd, cfg, m[:, :16])
            t0 = time.perf_counter()
            hifigan_cpu.synthesize(sd, cfg, m)
            r = a.cpu_frames / (time.perf_counter() - t0)
            if r > best:
                best, bt = r, nt
        line["cpu_baseline"] = {"value": best * cfg.hop, "unit": "samples/s", "cores": bt, "kind": "port",
                                "sample": f"oracle/hifigan_cpu.py (torch CPU fp32 conv1d/conv_transpose1d) on one utterance of "
                                          f"{a.cpu_frames} frames, best of a thread sweep"}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
