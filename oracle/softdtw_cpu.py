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


def _soft_dtw_R(D: np.ndarray, gamma: float) -> np.ndarray:
    """compute_softdtw :8-24: the whole (B, N+2, M+2) float64 table."""
    B, N, M = D.shape
    R = np.full((B, N + 2, M + 2), np.inf)
    R[:, 0, 0] = 0.0
    for d in range(2, N + M + 1):
        i = np.arange(max(1, d - M), min(N, d - 1) + 1)
        j = d - i
        r = -np.stack([R[:, i - 1, j - 1], R[:, i - 1, j], R[:, i, j - 1]]) / gamma
        rmax = r.max(axis=0)
        R[:, i, j] = D[:, i - 1, j - 1] - gamma * (np.log(np.exp(r - rmax).sum(axis=0)) + rmax)
    return R


def soft_dtw_value_and_grad(x, y, gamma: float = 1.0):
    """What ``SoftDTW(gamma)(x, y).sum().backward()`` leaves in ``x.grad`` through the reference's vendored module: the value
    (float32), E = compute_softdtw_backward (:27-52) evaluated in float64 on the FLOAT32 copies of D and R that
    ``_SoftDTW.forward`` saves (:63 ``torch.Tensor(R).type(dtype)``), handed back as float32 (:75), and the chain rule through
    ``calc_distance_matrix`` (:85-92): d/dx[i] = 2 sum_j E[i, j] (x[i] - y[j]).  One anti-diagonal at a time."""
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    B, N, _ = x.shape
    M = y.shape[1]
    D32 = distance_matrix(x, y)
    R = _soft_dtw_R(D32.astype(np.float64), gamma)
    value = R[:, N, M].astype(np.float32)
    R = R.astype(np.float32).astype(np.float64)      # the saved tensor is float32
    Dp = np.zeros((B, N + 2, M + 2))
    Dp[:, 1:N + 1, 1:M + 1] = D32
    E = np.zeros((B, N + 2, M + 2))
    E[:, -1, -1] = 1
    R[:, :, -1] = -np.inf
    R[:, -1, :] = -np.inf
    R[:, -1, -1] = R[:, -2, -2]
    with np.errstate(over="ignore", invalid="ignore"):
        for d in range(N + M, 1, -1):
            i = np.arange(max(1, d - M), min(N, d - 1) + 1)
            j = d - i
            a = np.exp((R[:, i + 1, j] - R[:, i, j] - Dp[:, i + 1, j]) / gamma)
            b = np.exp((R[:, i, j + 1] - R[:, i, j] - Dp[:, i, j + 1]) / gamma)
            c = np.exp((R[:, i + 1, j + 1] - R[:, i, j] - Dp[:, i + 1, j + 1]) / gamma)
            E[:, i, j] = E[:, i + 1, j] * a + E[:, i, j + 1] * b + E[:, i + 1, j + 1] * c
    E32 = E[:, 1:N + 1, 1:M + 1].astype(np.float32)
    # 2 * (x[i] * sum_j E[i,j] - sum_j E[i,j] y[j]), float32 like the autograd of pow(x - y, 2).sum(3)
    grad = 2.0 * (x * E32.sum(axis=2, dtype=np.float32)[:, :, None] - np.einsum("bij,bjd->bid", E32, y)).astype(np.float32)
    return value, grad.astype(np.float32)
