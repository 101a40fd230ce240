"""fp32 engine with every slab GEMM launch on bf16 x 3 split products (knob 501) against the exact fp32-MFMA engine: time per
forward at the C2 workload and the difference it makes (mel, decisions).  python tools/probes/split3_full.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from lightningfastspeech2_amd import _lib
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.weights import synth_state_dict, synth_inputs
from lightningfastspeech2_amd.model import FastSpeech2

cfg = preset("c2")
lib = _lib.load()
for name, skw, B, L, lengths in (("fixed-T", dict(duration_bias=float(np.log(7.0)), duration_weight_scale=0.0), 32, 256, None),
                                 ("ragged random heads", dict(duration_bias=1.5, randomize_norm=True), 8, 64, [64, 50, 33, 7, 64, 12, 40, 64])):
    sd = synth_state_dict(cfg, 0, **skw)
    inp = synth_inputs(cfg, B, L, seed=1234, lengths=lengths)
    m = FastSpeech2(cfg, sd, precision="fp32")
    batch = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}
    res = {}
    for knob in (500, 501):
        lib.fs2_op_set_gemm_variant(knob)
        for _ in range(2): out = m(batch, inference=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): out = m(batch, inference=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5 * 1e3
        res[knob] = ({k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, dt)
    lib.fs2_op_set_gemm_variant(500)
    a, b = res[500][0], res[501][0]
    same_T = a["mel"].shape == b["mel"].shape
    print(f"{name}: fp32 MFMA {res[500][1]:.2f} ms, bf16 x 3 split {res[501][1]:.2f} ms per forward; durations equal {bool(torch.equal(a['duration_rounded'], b['duration_rounded']))}"
          + (f", mel max-abs diff {float((a['mel'] - b['mel']).abs().max()):.3e} (scale {float(a['mel'].abs().max()):.2f})" if same_T else ", T differs")
          + "".join(f", {k} max diff {float((a[k] - b[k]).abs().max()):.2e}" for k in a if k.startswith("variances_") and same_T))
