#!/usr/bin/env python
"""Golden vectors for FastSpeech2Loss (forward half): runs the REFERENCE's own
`litfass.fastspeech2.loss.FastSpeech2Loss` (imported from /root/reference, build container only) on the
teacher-forced forward of tests/golden/teacher_small.npz (itself an output of the reference's
FastSpeech2.forward) plus seeded mel targets, for several loss configurations.

    python tools/gen_golden_loss.py      ->  tests/golden/loss_small.npz

The fixture holds the inputs (result tensors, targets) and the loss dicts the reference returned."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402

CASES = [
    dict(name="default", mel_loss="l1", duration_loss="mse", variance_losses=["mse", "mse", "mse"], alphas=None),
    dict(name="l2mel_l1dur", mel_loss="mse", duration_loss="l1", variance_losses=["l1", "mse", "l1"],
         alphas={"mel": 0.5, "pitch": 0.2, "energy": 0.3, "snr": 0.05, "duration": 1e-2, "fastdiff": 0.1, "speakers": 1}),
]


def main():
    assert ref_import.reference_available(), "needs /root/reference (build container only)"
    ref_import._install_stubs()
    sys.modules["pysdtw"].SoftDTW = lambda *a, **k: None  # constructed in __init__, used only by soft_dtw losses
    from litfass.fastspeech2.loss import FastSpeech2Loss

    z = np.load(os.path.join(ROOT, "tests", "golden", "teacher_small.npz"))
    cfg = json.loads(str(z["config_json"]))
    variances = cfg["variances"]
    result = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out_")}
    B, T, M = result["mel"].shape
    rs = np.random.RandomState(4242)
    target = {"mel": torch.from_numpy((rs.randn(B, T, M) * 1.3 - 2.0).astype(np.float32)),
              "duration": torch.from_numpy(z["tf_duration"])}
    for v in variances:
        target[f"variances_{v}"] = torch.from_numpy(z[f"tf_variances_{v}"])
    out = {"variances_json": json.dumps(variances), "cases_json": json.dumps(CASES)}
    for k, v in result.items():
        out["res_" + k] = v.numpy()
    for k, v in target.items():
        out["tgt_" + k] = v.numpy()
    for c in CASES:
        kw = dict(variances=variances, variance_levels=["frame"] * len(variances),
                  variance_transforms=["none"] * len(variances), variance_losses=c["variance_losses"],
                  mel_loss=c["mel_loss"], duration_loss=c["duration_loss"], max_length=4096)
        if c["alphas"]:
            kw["loss_alphas"] = dict(c["alphas"])
        lf = FastSpeech2Loss(**kw)
        losses = lf({k: v.clone() for k, v in result.items()}, {k: v.clone() for k, v in target.items()})
        for k, v in losses.items():
            out[f"loss_{c['name']}_{k}"] = np.float64(v.item())
        print(c["name"], {k: float(v) for k, v in losses.items()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_small.npz"), **out)


if __name__ == "__main__":
    main()
