"""Soft-DTW kernel (`-m gpu`) against the fixture the reference's vendored module produced and against the oracle at the
size the validation metric really runs at (full-length mels: the sequences no longer fit LDS and are read through the caches)."""
import json
import os

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.softdtw import SoftDTW
from oracle import softdtw_cpu

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "softdtw_small.npz"))
CASES = json.loads(str(Z["cases_json"]))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_matches_reference_fixture(case):
    n = case["name"]
    got = SoftDTW(gamma=case["gamma"], normalize=case["normalize"])(torch.from_numpy(Z[f"{n}__x"]), torch.from_numpy(Z[f"{n}__y"]))
    ref = Z[f"{n}__out"]
    assert tuple(got.shape) == ref.shape and got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=3e-6, atol=3e-4)


@pytest.mark.parametrize("N,M,D,gamma", [(700, 650, 80, 1.0), (1536, 1500, 80, 0.001), (240, 240, 80, 0.1), (1, 5, 3, 1.0)])
def test_long_sequences_vs_oracle(N, M, D, gamma):
    rs = np.random.RandomState(N + M)
    x, y = (rs.randn(2, N, D) * 0.7).astype(np.float32), (rs.randn(2, M, D) * 0.7 + 0.2).astype(np.float32)
    ref = softdtw_cpu.soft_dtw(x, y, gamma, normalize=False)
    got = SoftDTW(gamma=gamma)(torch.from_numpy(x), torch.from_numpy(y)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=5e-6)
    again = SoftDTW(gamma=gamma)(torch.from_numpy(x), torch.from_numpy(y)).cpu().numpy()
    assert np.array_equal(got, again)
