"""CPU: the training-step oracle (oracle/train_cpu.py: autograd over the forward oracle + AdamW/Noam/clipping) against the
fixture the REAL reference produced (tests/golden/train_small.npz, tools/gen_golden_train.py): losses, every parameter's
gradient, the clipped gradient norms, the Noam rates and every parameter after three optimizer steps."""
import json
import os

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.config import Fs2Config
from lightningfastspeech2_amd.weights import synth_state_dict
from oracle import train_cpu

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["train_small", "train_dw_small", "train_recipe_small"]  # dense family / the reference's depth-wise class defaults / the shipped recipe's architecture (scripts/train.sh)


def load(name="train_small"):
    z = np.load(os.path.join(GOLD_DIR, f"{name}.npz"))
    cfg = Fs2Config.from_json(str(z["config_json"]))
    skw = json.loads(str(z["synth_json"]))
    sd = synth_state_dict(cfg, skw.pop("seed"), **skw)
    batch = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    return z, cfg, sd, batch, json.loads(str(z["hyper_json"]))


def assert_params_close(z, name, got, tol):
    """Adam divides by sqrt(v): where a gradient is numerical noise around zero (the key bias of an attention layer has an
    exactly-zero derivative - softmax is shift invariant) the update is +-lr whatever the noise was, so such entries are only
    bounded by the three steps' learning rates; everywhere else the weights must agree to `tol`."""
    want, g1 = torch.from_numpy(z["after3_" + name]), torch.from_numpy(z["grad_" + name]).abs()
    diff = (got.cpu().float() - want).abs()
    solid = g1 > 1e-5 * max(1.0, float(g1.max()))
    assert float(diff[solid].max() if solid.any() else 0.0) <= tol, (name, float(diff[solid].max()))
    assert float(diff.max()) <= 2.5e-3, (name, float(diff.max()))


@pytest.mark.parametrize("name", CASES)
def test_oracle_training_matches_reference_fixture(name):
    z, cfg, sd, batch, hyper = load(name)
    tr = train_cpu.OracleTrainer(cfg, sd, **hyper)
    for step in (1, 2, 3):
        ls, _ = tr.training_step(batch)
        if step == 1:
            for k, v in ls.items():
                assert abs(v - float(z[f"loss_{k}"])) <= 1e-5 * max(1.0, abs(float(z[f"loss_{k}"]))), k
            grads = tr.gradients()
            names = [k[5:] for k in z.files if k.startswith("grad_")]
            assert sorted(names) == sorted(grads)
            for n in names:
                want = torch.from_numpy(z["grad_" + n])
                err = float((grads[n] - want).abs().max())
                assert err <= 2e-5 * (float(want.abs().max()) + 1e-3), (n, err)
        norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in tr.params if p.grad is not None)))
        assert abs(norm - float(z[f"gradnorm_{step}"])) <= 1e-4 * float(z[f"gradnorm_{step}"])
        lr = tr.optimizer_step()
        assert abs(lr - float(z[f"lr_{step}"])) <= 1e-12
    for k in z.files:
        if k.startswith("after3_"):
            assert_params_close(z, k[7:], tr.sd[k[7:]].detach(), 5e-6)
