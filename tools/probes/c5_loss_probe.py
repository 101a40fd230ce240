import sys, math, numpy as np, torch
sys.path[:0] = [".", "tests"]
from test_gpu_training import synth_state_dict, synth_inputs, _dev
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.training import Trainer
cfg = preset("c5")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(4.0), duration_weight_scale=0.0)
B, L, f = 2, 48, 3
inp = synth_inputs(cfg, B, L, seed=77, lengths=[L, L - 11])
rs = np.random.RandomState(3)
dur = np.full((B, L), f, np.int64); dur[1, L - 11:] = 0
T = L * f
batch = {"phones": inp["phones"], "speaker": inp["speaker"], "duration": dur, "mel": (rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)}
for v in cfg.variances: batch[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
bd = _dev(batch)
for lr in (1e-3, 1e-4, 2e-5):
    for prec in ("bf16", "fp32"):
        tr = Trainer(cfg, sd, precision=prec, lr=lr, warmup_steps=1)
        out = []
        for _ in range(5):
            l = tr.training_step(bd); out.append(round(float(l["total"]), 4)); tr.optimizer_step()
        print(lr, prec, out)
