import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_training import _case, _dev
from lightningfastspeech2_amd.training import Trainer
cfg, sd, batch = _case(31, 3, 12, [12, 7, 3])
bd = _dev(batch)
def check(name, **drop):
    tr = Trainer(cfg, sd, gradient_clip_val=None, seed=9, **drop)
    l0 = float(tr.training_step(bd)["total"])
    g = tr.flat_g.clone().double(); tr.zero_grad()
    w0 = tr.flat_p.clone()
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    res = []
    for trial in range(2):
        v = torch.randn(tr.n_flat, device="cuda:0", generator=gen) * (w0.abs() + 1e-2)
        for eps in (2e-3, 5e-4):
            vals = []
            for sgn in (1.0, -1.0):
                tr.flat_p.copy_(w0 + sgn * eps * v); tr._refresh_shadow(); tr._micro = 0
                vals.append(float(tr.training_step(bd)["total"].double())); tr.zero_grad()
            res.append(((vals[0] - vals[1]) / (2 * eps), float((g * v.double()).sum())))
    print(name, l0, " ".join(f"fd={a:.5f}/an={b:.5f}" for a, b in res))
check("none")
check("enc", encoder_dropout=0.2)
check("dec", decoder_dropout=0.2)
check("var", variance_dropout=0.3)
check("dur", duration_dropout=0.3)
