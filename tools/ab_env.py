#!/usr/bin/env python
"""A/B of an engine-creation env knob on the C2 bench workload, interleaved rounds in ONE process:
    python tools/ab_env.py FS2_PRED_WCOPIES 1 4 [config]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
var, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = preset(sys.argv[4] if len(sys.argv) > 4 else "c2")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
models = {}
for v in (a, b):
    os.environ[var] = v
    models[v] = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
bt = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}
def run(m, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(bt, inference=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for m in models.values(): run(m, 3)
res = {a: [], b: []}
for r in range(7):
    for v in (a, b): res[v].append(run(models[v], 10))
oa, ob = models[a](bt, inference=True), models[b](bt, inference=True)
print("bit-equal mel:", bool(torch.equal(oa["mel"], ob["mel"])))
for v in (a, b): print(f"{var}={v}", " ".join(f"{x:.3f}" for x in res[v]), "median %.3f ms" % sorted(res[v])[3])
