"""Loader for tests/golden/*.npz (written by tools/gen_golden.py from the real reference)."""
import glob
import hashlib
import json
import os

import numpy as np

from lightningfastspeech2_amd.config import Fs2Config
from lightningfastspeech2_amd.weights import synth_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    # vocoder / front-end fixtures have their own tests (test_hifigan_oracle.py, test_frontend_cpu.py)
    return [n for n in names if not n.startswith(("hifigan_", "frontend_", "loss_", "softdtw_", "train_"))]


def sd_digest(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        if k.endswith(".pe") or k.endswith(".bins"):
            continue  # libm-dependent in the last ulp across hosts; they travel inside the state_dict
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        self.name = name
        self.cfg = Fs2Config.from_json(str(z["config_json"]))
        self.synth = json.loads(str(z["synth_json"]))
        self.sd_sha256 = str(z["sd_sha256"])
        self.phones = z["phones"]
        self.speaker = z["speaker"]
        self.priors = {k[3:]: z[k] for k in z.files if k.startswith("in_priors_")}
        self.teacher = {k[3:]: z[k] for k in z.files if k.startswith("tf_")} or None
        self.out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
        self.mid = {k[4:]: z[k] for k in z.files if k.startswith("mid_")}
        self.margins = z["margins"]
        self.n_guard = int(z["n_guard"])

    def state_dict(self):
        sd = synth_state_dict(self.cfg, **self.synth)
        assert sd_digest(sd) == self.sd_sha256, "synthetic weight recipe drifted from the fixture"
        return sd


def lookup(out, key):
    """out["variances_pitch.mean"] -> out["variances_pitch"]["mean"]: the CWT head returns a dict per variance."""
    if "." in key:
        a, b = key.split(".", 1)
        return out[a][b]
    return out[key]
