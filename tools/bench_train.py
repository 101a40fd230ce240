#!/usr/bin/env python
"""Training-step timing on synthetic data: teacher-forced forward + losses + backward + optimizer step.
    python tools/bench_train.py [--config c2] [--batch 32] [--phones 256] [--steps 5] [--warmup 2]
Prints one JSON line: ms per step, mel frames per second through the training step, and the split forward / backward."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningfastspeech2_amd.config import preset  # noqa: E402
from lightningfastspeech2_amd.training import Trainer  # noqa: E402
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--phones", type=int, default=256)
    ap.add_argument("--frames-per-phone", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--dropout", type=float, default=0.0, help="every nn.Dropout site of the reference at this rate (the shipped recipe: 0.1)")
    ap.add_argument("--knob", type=int, action="append", default=[], help="fs2_op_set_gemm_variant values (A/B switches)")
    a = ap.parse_args()
    for k in a.knob:
        from lightningfastspeech2_amd import _lib
        _lib.load().fs2_op_set_gemm_variant(k)
    cfg = preset(a.config)
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
    B, L = a.batch, a.phones
    T = L * a.frames_per_phone
    inp = synth_inputs(cfg, B, L, seed=1234)
    rs = np.random.RandomState(5)
    batch = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda(),
             "duration": torch.full((B, L), a.frames_per_phone, dtype=torch.int64).cuda(),
             "mel": torch.from_numpy((rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)).cuda()}
    for v in cfg.variances:
        batch[f"variances_{v}"] = torch.from_numpy(rs.randn(B, T).astype(np.float32)).cuda()
    kw = {} if a.precision == "fp32" else {"precision": a.precision}
    if a.dropout > 0:
        kw.update(encoder_dropout=a.dropout, decoder_dropout=a.dropout, variance_dropout=a.dropout, duration_dropout=a.dropout)
    tr = Trainer(cfg, sd, **kw)
    for _ in range(a.warmup):
        losses = tr.training_step(batch)
        tr.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = tr.training_step(batch)
        tr.optimizer_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "training step (forward + loss + backward + AdamW)", "config": a.config, "batch": B, "phones": L,
                      "frames": T, "precision": a.precision, "ms_per_step": dt * 1e3, "mel_frames_per_s": B * T / dt,
                      "loss_total": float(losses["total"]), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == "__main__":
    main()
