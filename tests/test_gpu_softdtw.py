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


ZG = np.load(os.path.join(os.path.dirname(__file__), "golden", "softdtw_grad_small.npz"))
GRAD_CASES = json.loads(str(ZG["cases_json"]))


@pytest.mark.parametrize("case", GRAD_CASES, ids=[c["name"] for c in GRAD_CASES])
def test_gradient_matches_reference_backward(case):
    """fs2_op_soft_dtw_grad against x.grad of the reference's own autograd path (vendored _SoftDTW.backward +
    calc_distance_matrix), tools/gen_golden_softdtw.py."""
    from lightningfastspeech2_amd.softdtw import soft_dtw_value_and_grad
    n = case["name"]
    x, y = torch.from_numpy(ZG[f"{n}__x"]).cuda(), torch.from_numpy(ZG[f"{n}__y"]).cuda()
    val, grad = soft_dtw_value_and_grad(x, y, case["gamma"])
    np.testing.assert_allclose(val.cpu().numpy(), ZG[f"{n}__out"], rtol=3e-6, atol=3e-4)
    want = ZG[f"{n}__grad"]
    assert float(np.abs(grad.cpu().numpy() - want).max()) <= 1e-3 * float(np.abs(want).max())
    val2, grad2 = soft_dtw_value_and_grad(x, y, case["gamma"])
    assert torch.equal(grad, grad2) and torch.equal(val, val2)


@pytest.mark.parametrize("N,M,D,gamma", [(256, 256, 80, 1.0), (256, 200, 1, 0.5), (100, 256, 80, 2.0)])
def test_gradient_at_the_loss_chunk_size_vs_oracle(N, M, D, gamma):
    """The size the "soft_dtw" loss kind runs at (loss.py:62-81: chunks of soft_dtw_chunk_size = 256 frames of an 80-bin mel, or
    of a 1-channel variance): sequences read through the caches, value and gradient against the oracle; and the gradient is the
    directional derivative of the value (central differences in float64 on the oracle's own value)."""
    from lightningfastspeech2_amd.softdtw import soft_dtw_value_and_grad
    rs = np.random.RandomState(N + M + D)
    x, y = (rs.randn(2, N, D) * 0.7).astype(np.float32), (rs.randn(2, M, D) * 0.7 + 0.2).astype(np.float32)
    val, grad = soft_dtw_value_and_grad(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), gamma)
    rv, rg = softdtw_cpu.soft_dtw_value_and_grad(x, y, gamma)
    np.testing.assert_allclose(val.cpu().numpy(), rv, rtol=5e-6)
    g = grad.cpu().numpy()
    assert float(np.abs(g - rg).max()) <= 2e-3 * float(np.abs(rg).max())
    v = rs.randn(*x.shape).astype(np.float32)
    eps = 1e-2
    up = softdtw_cpu.soft_dtw(x + eps * v, y, gamma).astype(np.float64)
    dn = softdtw_cpu.soft_dtw(x - eps * v, y, gamma).astype(np.float64)
    fd = (up - dn) / (2 * eps)
    an = (g.astype(np.float64) * v).sum(axis=(1, 2))
    # a directional derivative is a sum of terms of both signs: tolerance relative to the terms' total size
    np.testing.assert_allclose(an, fd, rtol=2e-2, atol=2e-3 * float(np.abs(g.astype(np.float64) * v).sum(axis=(1, 2)).max()))
