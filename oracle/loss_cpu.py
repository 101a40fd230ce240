"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's
``FastSpeech2Loss.forward`` (litfass/fastspeech2/loss.py:83-213) for the configuration the shipped recipe
trains with - frame-level variances, transform "none", losses "l1" / "mse", no FastDiff, deterministic
durations - in plain numpy, float64 accumulation.

Pinned by tests/golden/loss_small.npz = the loss dicts the reference's own class returned on the
teacher-forced forward of tests/golden/teacher_small.npz (tools/gen_golden_loss.py).

Follows, line by line:
  get_loss (loss.py:57-81)  : masked_select(pred), masked_select(truth) with the NON-pad mask, then
                              nn.L1Loss() / nn.MSELoss() = mean over the selected elements
  variances (loss.py:99-158): pred result["variances_{v}"] (B,T) vs target[:, :max_length], mask ~tgt_mask
  mel       (loss.py:161-167): (B,T,n_mels), mask ~tgt_mask broadcast over the bins
  duration  (loss.py:179-185): result["duration_prediction"] vs log(target["duration"] + 1), mask ~src_mask
  total     (loss.py:203-211): sum of loss_alphas[k] * loss_k
"""
import numpy as np

DEFAULT_ALPHAS = {"mel": 1.0, "pitch": 1e-1, "energy": 1e-1, "snr": 1e-1, "duration": 1e-4, "fastdiff": 1e-1,
                  "speakers": 1}  # loss.py:19-27


def _get_loss(pred, truth, kind, keep):
    """loss.py:57-81 for kind in ("l1", "mse"); keep = boolean array broadcastable to pred (True = counted)."""
    keep = np.broadcast_to(keep, pred.shape)
    d = pred.astype(np.float64)[keep] - truth.astype(np.float64)[keep]
    if kind == "l1":
        return float(np.abs(d).mean()) if d.size else float("nan")
    if kind == "mse":
        return float((d * d).mean()) if d.size else float("nan")
    raise NotImplementedError(f"loss {kind!r} (soft_dtw is outside the restated configuration)")


def fastspeech2_loss(result, target, variances, variance_losses=None, mel_loss="l1", duration_loss="mse",
                     loss_alphas=None, max_length=4096):
    """result / target: dicts of numpy arrays named as in the reference; returns {name: float, "total": float}."""
    alphas = dict(DEFAULT_ALPHAS if loss_alphas is None else loss_alphas)
    variance_losses = variance_losses or ["mse"] * len(variances)
    src_keep = ~np.asarray(result["src_mask"], dtype=bool)
    tgt_keep = ~np.asarray(result["tgt_mask"], dtype=bool)
    losses = {}
    assert target["mel"].shape[1] <= max_length  # loss.py:101
    for v, kind in zip(variances, variance_losses):
        tgt = np.asarray(target[f"variances_{v}"])[:, :int(max_length)]
        losses[v] = _get_loss(np.asarray(result[f"variances_{v}"]), tgt.astype(np.float32), kind, tgt_keep)
    losses["mel"] = _get_loss(np.asarray(result["mel"]), np.asarray(target["mel"], dtype=np.float32), mel_loss,
                              tgt_keep[..., None])
    logd = np.log(np.asarray(target["duration"]).astype(np.float32) + np.float32(1.0)).astype(np.float32)
    losses["duration"] = _get_loss(np.asarray(result["duration_prediction"]), logd, duration_loss, src_keep)
    losses["total"] = float(sum(val * alphas[k] for k, val in losses.items()))
    return losses
