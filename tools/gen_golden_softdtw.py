#!/usr/bin/env python
"""Golden vectors for soft-DTW (the validation metric of fastspeech2.py:1149-1156 and the "soft_dtw" loss kind
of loss.py:57-81): runs the REFERENCE's own vendored litfass/third_party/softdtw/__init__.py (imported from /root/reference;
its numba.jit decorator is a pass-through stub here, so the recursion runs as the plain Python/numpy it is written in).

    python tools/gen_golden_softdtw.py      ->  tests/golden/softdtw_small.npz  (inputs + the values the reference returned)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402

CASES = [dict(name="g1_norm", gamma=1.0, normalize=True, B=3, N=19, M=23, D=8),
         dict(name="g0001_norm", gamma=0.001, normalize=True, B=2, N=17, M=17, D=80),
         dict(name="g01_plain", gamma=0.1, normalize=False, B=2, N=30, M=11, D=5),
         dict(name="unbatched", gamma=1.0, normalize=True, B=0, N=13, M=9, D=80)]


GRAD_CASES = [dict(name="grad_g1", gamma=1.0, B=3, N=19, M=23, D=8),
              dict(name="grad_g01", gamma=0.1, B=2, N=30, M=11, D=5),
              dict(name="grad_g05_mel", gamma=0.5, B=2, N=24, M=24, D=80),
              dict(name="grad_one_frame", gamma=1.0, B=1, N=1, M=6, D=4)]


def main():
    assert ref_import.reference_available()
    ref_import._install_stubs()
    from litfass.third_party.softdtw import SoftDTW
    out = {"cases_json": json.dumps(CASES)}
    for i, c in enumerate(CASES):
        rs = np.random.RandomState(900 + i)
        shp = lambda n: ((c["B"], n, c["D"]) if c["B"] else (n, c["D"]))
        x = (rs.randn(*shp(c["N"])) * 0.8 - 1.0).astype(np.float32)
        y = (rs.randn(*shp(c["M"])) * 0.8 - 1.0).astype(np.float32)
        val = SoftDTW(gamma=c["gamma"], normalize=c["normalize"])(torch.from_numpy(x), torch.from_numpy(y))
        out[f"{c['name']}__x"], out[f"{c['name']}__y"] = x, y
        out[f"{c['name']}__out"] = np.asarray(val.numpy(), dtype=np.float64)
        print(c["name"], np.asarray(val))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "softdtw_small.npz"), **out)
    # gradients: SoftDTW(gamma)(x, y).sum().backward() through the reference's own autograd Function (_SoftDTW.backward ->
    # compute_softdtw_backward) and calc_distance_matrix -> tests/golden/softdtw_grad_small.npz
    gout = {"cases_json": json.dumps(GRAD_CASES)}
    for i, c in enumerate(GRAD_CASES):
        rs = np.random.RandomState(950 + i)
        x = (rs.randn(c["B"], c["N"], c["D"]) * 0.8 - 1.0).astype(np.float32)
        y = (rs.randn(c["B"], c["M"], c["D"]) * 0.8 - 1.0).astype(np.float32)
        xt = torch.from_numpy(x).requires_grad_(True)
        val = SoftDTW(gamma=c["gamma"], normalize=False)(xt, torch.from_numpy(y))
        val.sum().backward()
        gout[f"{c['name']}__x"], gout[f"{c['name']}__y"] = x, y
        gout[f"{c['name']}__out"] = val.detach().numpy().astype(np.float64)
        gout[f"{c['name']}__grad"] = xt.grad.numpy().astype(np.float32)
        print(c["name"], val.detach().numpy(), float(xt.grad.abs().max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "softdtw_grad_small.npz"), **gout)


if __name__ == "__main__":
    main()
