"""CPU restatement of soft-DTW as the reference computes it (test infrastructure only; never imported by the product).

Follows /root/reference/litfass/third_party/softdtw/__init__.py: ``calc_distance_matrix`` :110-117 (squared Euclidean distance
of every frame pair, float32), ``compute_softdtw`` :8-24 (float64 recursion R[i,j] = D[i-1,j-1] + softmin_gamma(R[i-1,j-1],
R[i-1,j], R[i,j-1]) with R[0,0] = 0 and an infinite border, value = R[N,M], handed back as float32) and ``SoftDTW.forward``
:119-139 (``normalize``: out_xy - (out_xx + out_yy) / 2).  Pinned by tests/golden/softdtw_small.npz = what that module itself
returned (tools/gen_golden_softdtw.py).  The recursion is evaluated one anti-diagonal at a time."""
import numpy as np


def _soft_dtw_value(D: np.ndarray, gamma: float) -> np.ndarray:
    B, N, M = D.shape
    R = np.full((B, N + 2, M + 2), np.inf)
    R[:, 0, 0] = 0.0
    for d in range(2, N + M + 1):
        i = np.arange(max(1, d - M), min(N, d - 1) + 1)
        j = d - i
        r = -np.stack([R[:, i - 1, j - 1], R[:, i - 1, j], R[:, i, j - 1]]) / gamma
        rmax = r.max(axis=0)
        softmin = -gamma * (np.log(np.exp(r - rmax).sum(axis=0)) + rmax)
        R[:, i, j] = D[:, i - 1, j - 1] + softmin
    return R[:, N, M].astype(np.float32)


def distance_matrix(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    return ((x[:, :, None, :].astype(np.float32) - y[:, None, :, :].astype(np.float32)) ** 2).sum(-1, dtype=np.float32)


def soft_dtw(x, y, gamma: float = 1.0, normalize: bool = False) -> np.ndarray:
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    squeeze = x.ndim < 3
    if squeeze:
        x, y = x[None], y[None]
    out = _soft_dtw_value(distance_matrix(x, y).astype(np.float64), gamma)
    if normalize:
        out = out - np.float32(0.5) * (_soft_dtw_value(distance_matrix(x, x).astype(np.float64), gamma)
                                       + _soft_dtw_value(distance_matrix(y, y).astype(np.float64), gamma))
    return out[0] if squeeze else out
