import sys, math, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.training import Trainer
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = preset(name)
sd = synth_state_dict(cfg, 0, duration_bias=math.log(4.0), duration_weight_scale=0.0)
B, L, f = 4, 256, 3
lens = [L, L - 37, L - 90, L // 2]
inp = synth_inputs(cfg, B, L, seed=91, lengths=lens)
rs = np.random.RandomState(5)
dur = np.zeros((B, L), np.int64)
for b, n in enumerate(lens): dur[b, :n] = rs.randint(1, 2 * f, size=n)
T = int(dur.sum(axis=1).max())
batch = {"phones": inp["phones"], "speaker": inp["speaker"], "duration": dur, "mel": (rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)}
for v in cfg.variances: batch[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
bd = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
import os
kw = dict(lr=1e-3, warmup_steps=1, seed=11, attention=os.environ.get("ATT", "auto"))
ref = Trainer(cfg, sd, precision="fp32", **kw); wl = ref.training_step(bd); want = {n: g.double().cpu() for n, g in ref.gradients().items()}
tr = Trainer(cfg, sd, precision="bf16", **kw); gl = tr.training_step(bd); got = tr.gradients()
print({k: (round(float(wl[k]), 5), round(float(gl[k]), 5)) for k in wl})
gmax = max(float(w.abs().max()) for w in want.values())
rows = []
for n, w in want.items():
    g = got[n].double().cpu()
    cos = float((g * w).sum() / (g.norm() * w.norm() + 1e-30))
    rows.append((cos, n, float(w.abs().max()) / gmax, float(w.norm())))
rows.sort()
rows = [r for r in rows if r[3] > 0]
print("below 0.99:", sum(1 for r in rows if r[0] < 0.99), "of", len(rows), "min", rows[0][0])
for r in rows[:4]: print(f"{r[0]:.5f}  rel-max {r[2]:.2e}  norm {r[3]:.3e}  {r[1]}")
