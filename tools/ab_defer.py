#!/usr/bin/env python
"""Interleaved A/B of the deferred-LayerNorm path on C3 (LS-76M, B=32): same process, alternating rounds."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
cfg = preset(sys.argv[1] if len(sys.argv) > 1 else "c3")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
m = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
b = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(b, inference=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for on in (1, 0): m.engine.set_deferred_layernorm(bool(on)); run(3)
res = {0: [], 1: []}
for r in range(5):
    for on in (0, 1):
        m.engine.set_deferred_layernorm(bool(on)); res[on].append(run(10))
for on in (0, 1): print("deferred" if on else "launches", " ".join(f"{x:.3f}" for x in res[on]), "median %.3f ms" % sorted(res[on])[2])
