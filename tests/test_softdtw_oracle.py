"""The soft-DTW oracle against what the reference's vendored third_party/softdtw module returned (tests/golden/softdtw_small.npz)."""
import json
import os

import numpy as np
import pytest

from oracle import softdtw_cpu

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "softdtw_small.npz"))
CASES = json.loads(str(Z["cases_json"]))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_fixture(case):
    n = case["name"]
    got = softdtw_cpu.soft_dtw(Z[f"{n}__x"], Z[f"{n}__y"], gamma=case["gamma"], normalize=case["normalize"])
    ref = Z[f"{n}__out"]
    assert np.shape(got) == ref.shape
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-4)


def test_identical_sequences_normalise_to_zero_and_plain_value_is_a_soft_minimum():
    rs = np.random.RandomState(0)
    x = rs.randn(2, 12, 6).astype(np.float32)
    assert np.abs(softdtw_cpu.soft_dtw(x, x, 0.5, True)).max() <= 1e-4
    y = rs.randn(2, 9, 6).astype(np.float32)
    hard = softdtw_cpu.soft_dtw(x, y, 1e-4)          # gamma -> 0: the DTW alignment cost
    soft = softdtw_cpu.soft_dtw(x, y, 1.0)
    assert (soft <= hard + 1e-3).all()                # a soft minimum never exceeds the hard one


def test_gradient_oracle_matches_the_reference_backward():
    """oracle.softdtw_cpu.soft_dtw_value_and_grad against what loss.backward() left in x.grad through the reference's own
    _SoftDTW Function + calc_distance_matrix (tools/gen_golden_softdtw.py -> tests/golden/softdtw_grad_small.npz)."""
    import json
    import os
    zg = np.load(os.path.join(os.path.dirname(__file__), "golden", "softdtw_grad_small.npz"))
    for c in json.loads(str(zg["cases_json"])):
        n = c["name"]
        val, grad = softdtw_cpu.soft_dtw_value_and_grad(zg[f"{n}__x"], zg[f"{n}__y"], c["gamma"])
        np.testing.assert_allclose(val, zg[f"{n}__out"], rtol=3e-6, atol=3e-4)
        want = zg[f"{n}__grad"]
        # the backward exponentiates float32-rounded R differences divided by gamma: last-ulp differences in D move E by ~1e-4
        assert float(np.abs(grad - want).max()) <= 1e-3 * float(np.abs(want).max()), n
