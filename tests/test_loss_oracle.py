"""CPU: the loss oracle (oracle/loss_cpu.py) against the loss dicts the reference's own FastSpeech2Loss
returned (tests/golden/loss_small.npz, tools/gen_golden_loss.py)."""
import json
import os

import numpy as np
import pytest

from oracle import loss_cpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_small.npz")
REL = 2e-6  # fp32 reference means vs float64 restatement


def load():
    z = np.load(GOLD)
    res = {k[4:]: z[k] for k in z.files if k.startswith("res_")}
    tgt = {k[4:]: z[k] for k in z.files if k.startswith("tgt_")}
    return z, res, tgt, json.loads(str(z["variances_json"])), json.loads(str(z["cases_json"]))


@pytest.mark.parametrize("case", [0, 1])
def test_oracle_matches_reference_losses(case):
    z, res, tgt, variances, cases = load()
    c = cases[case]
    got = loss_cpu.fastspeech2_loss(res, tgt, variances, c["variance_losses"], c["mel_loss"], c["duration_loss"], c["alphas"])
    keys = variances + ["mel", "duration", "total"]
    for k in keys:
        want = float(z[f"loss_{c['name']}_{k}"])
        assert abs(got[k] - want) <= REL * max(1.0, abs(want)), (k, got[k], want)


def test_oracle_ignores_pad_rows():
    z, res, tgt, variances, cases = load()
    base = loss_cpu.fastspeech2_loss(res, tgt, variances)
    res2 = {k: v.copy() for k, v in res.items()}
    pad = res["tgt_mask"].astype(bool)
    assert pad.any(), "fixture must contain padded frames"
    res2["mel"][pad] += 100.0
    for v in variances:
        res2[f"variances_{v}"][pad] -= 50.0
    res2["duration_prediction"][res["src_mask"].astype(bool)] = 1e6
    assert loss_cpu.fastspeech2_loss(res2, tgt, variances) == base


def test_unsupported_kind_is_loud():
    z, res, tgt, variances, cases = load()
    with pytest.raises(NotImplementedError):
        loss_cpu.fastspeech2_loss(res, tgt, variances, mel_loss="soft_dtw")
