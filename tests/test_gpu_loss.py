"""GPU: FastSpeech2Loss (lightningfastspeech2_amd/loss.py -> fs2_op_masked_loss, csrc/loss.hip) against the
reference's loss values (tests/golden/loss_small.npz) and against the oracle at full size.
Tolerance: 1e-5 relative (fp32 pred/truth, fp64 accumulation on both sides; the reference sums in fp32)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import loss_cpu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_small.npz")
REL = 1e-5


def _cuda(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}


def _load():
    z = np.load(GOLD)
    res = {k[4:]: z[k] for k in z.files if k.startswith("res_")}
    tgt = {k[4:]: z[k] for k in z.files if k.startswith("tgt_")}
    return z, res, tgt, json.loads(str(z["variances_json"])), json.loads(str(z["cases_json"]))


@pytest.mark.parametrize("case", [0, 1])
def test_loss_matches_reference_fixture(case):
    from lightningfastspeech2_amd.loss import FastSpeech2Loss
    z, res, tgt, variances, cases = _load()
    c = cases[case]
    kw = dict(variances=variances, variance_levels=["frame"] * 3, variance_transforms=["none"] * 3,
              variance_losses=c["variance_losses"], mel_loss=c["mel_loss"], duration_loss=c["duration_loss"])
    if c["alphas"]:
        kw["loss_alphas"] = c["alphas"]
    got = FastSpeech2Loss(**kw)(_cuda(res), _cuda(tgt))
    for k in variances + ["mel", "duration", "total"]:
        want = float(z[f"loss_{c['name']}_{k}"])
        assert abs(float(got[k]) - want) <= REL * max(1.0, abs(want)), (k, float(got[k]), want)


def _synthetic(B, L, T, M, seed):
    rs = np.random.RandomState(seed)
    src_len = rs.randint(L // 2, L + 1, size=B); src_len[0] = L
    tgt_len = rs.randint(T // 2, T + 1, size=B); tgt_len[0] = T
    res = {"mel": rs.randn(B, T, M).astype(np.float32), "duration_prediction": rs.rand(B, L).astype(np.float32) * 2,
           "src_mask": np.arange(L)[None, :] >= src_len[:, None], "tgt_mask": np.arange(T)[None, :] >= tgt_len[:, None]}
    tgt = {"mel": (rs.randn(B, T, M) * 1.5 - 3).astype(np.float32), "duration": rs.randint(0, 12, size=(B, L)).astype(np.int64)}
    for v in ("pitch", "energy", "snr"):
        res[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
        tgt[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
    return res, tgt


@pytest.mark.parametrize("shape", [(32, 256, 1536, 80), (3, 7, 19, 8), (1, 1, 1, 1), (5, 33, 2756, 80)])
def test_loss_matches_oracle_at_size(shape):
    from lightningfastspeech2_amd.loss import FastSpeech2Loss
    res, tgt = _synthetic(*shape, seed=sum(shape))
    variances = ["pitch", "energy", "snr"]
    want = loss_cpu.fastspeech2_loss(res, tgt, variances, ["mse", "l1", "mse"], "l1", "mse")
    lf = FastSpeech2Loss(variances=variances, variance_losses=["mse", "l1", "mse"])
    a = lf(_cuda(res), _cuda(tgt))
    b = lf(_cuda(res), _cuda(tgt))
    for k, w in want.items():
        assert abs(float(a[k]) - w) <= REL * max(1.0, abs(w)), (k, float(a[k]), w)
        assert float(a[k]) == float(b[k]), "the reduction must be deterministic"


def test_loss_pad_rows_do_not_count_and_empty_is_nan():
    from lightningfastspeech2_amd.loss import FastSpeech2Loss
    res, tgt = _synthetic(4, 16, 64, 80, seed=5)
    lf = FastSpeech2Loss(variances=["pitch", "energy", "snr"])
    base = {k: float(v) for k, v in lf(_cuda(res), _cuda(tgt)).items()}
    res2 = {k: v.copy() for k, v in res.items()}
    res2["mel"][res["tgt_mask"]] = 1e9
    res2["duration_prediction"][res["src_mask"]] = -1e9
    again = {k: float(v) for k, v in lf(_cuda(res2), _cuda(tgt)).items()}
    assert again == base
    res3 = {k: v.copy() for k, v in res.items()}
    res3["tgt_mask"][:] = True  # nothing selected: torch's mean of an empty selection is nan
    assert np.isnan(float(lf(_cuda(res3), _cuda(tgt))["mel"]))


def test_loss_rejects_configurations_outside_the_path():
    from lightningfastspeech2_amd.loss import FastSpeech2Loss
    for kw in (dict(mel_loss="huber"), dict(variance_transforms=["cwt", "none", "none"]),
               dict(variance_levels=["phone"] * 3), dict(duration_stochastic=True), dict(fastdiff_loss="mse")):
        with pytest.raises(NotImplementedError):
            FastSpeech2Loss(**kw)


def test_validation_step_forward_plus_loss():
    """fastspeech2.py:800-802: result = self(batch); losses = self.loss(result, batch) - teacher-forced forward of the
    reference fixture through the engine (fp32 mode), then the loss; against the oracle loss of the REFERENCE's outputs."""
    from _golden import Golden
    from lightningfastspeech2_amd.loss import FastSpeech2Loss
    from lightningfastspeech2_amd.model import FastSpeech2
    g = Golden("teacher_small")
    m = FastSpeech2(g.cfg, g.state_dict(), precision="fp32", device="cuda:0")
    batch = {"phones": torch.from_numpy(g.phones), "speaker": torch.from_numpy(g.speaker)}
    batch.update({k: torch.from_numpy(v) for k, v in g.teacher.items()})
    rs = np.random.RandomState(99)
    batch["mel"] = torch.from_numpy((rs.randn(*g.out["mel"].shape) - 2).astype(np.float32))
    out = m(batch)  # inference=False
    got = FastSpeech2Loss(variances=g.cfg.variances)(out, {k: v.cuda() for k, v in batch.items() if k != "phones" and k != "speaker"})
    ref_res = dict(g.out)
    want = loss_cpu.fastspeech2_loss(ref_res, {k: v.numpy() for k, v in batch.items()}, g.cfg.variances)
    for k, w in want.items():
        assert abs(float(got[k]) - w) <= 2e-4 * max(1.0, abs(w)), (k, float(got[k]), w)  # mel parity 1e-3 abs -> loss


def test_soft_dtw_loss_kind_chunked():
    """mel_loss = "soft_dtw" (loss.py:60-78): zero-filled pads, chunks of soft_dtw_chunk_size frames, soft-DTW value of every
    chunk pair summed over chunks and batch - against the oracle's soft-DTW (pinned on the reference's vendored module)."""
    from lightningfastspeech2_amd.loss import FastSpeech2Loss
    from oracle import softdtw_cpu
    res, tgt = _synthetic(3, 9, 150, 80, seed=77)
    gamma, chunk = 0.05, 64
    loss = FastSpeech2Loss(variance_levels=["frame"] * 3, variance_transforms=["none"] * 3, variance_losses=["mse", "soft_dtw", "mse"],
                           mel_loss="soft_dtw", soft_dtw_gamma=gamma, soft_dtw_chunk_size=chunk)
    got = loss(_cuda(res), _cuda(tgt))
    valid = ~res["tgt_mask"]

    def want(pred, truth):
        if pred.ndim == 2:
            pred, truth = pred[..., None], truth[..., None]
        p, t = pred * valid[..., None], truth * valid[..., None]
        tot = 0.0
        for s in range(0, p.shape[1], chunk):
            tot += float(softdtw_cpu.soft_dtw(p[:, s:s + chunk], t[:, s:s + chunk], gamma).astype(np.float64).sum())
        return tot
    for key, (pr, tr) in {"mel": (res["mel"], tgt["mel"]), "pitch": (res["variances_pitch"], tgt["variances_pitch"])}.items():
        w = want(pr, tr)
        assert abs(float(got[key]) - w) <= 2e-5 * abs(w), (key, float(got[key]), w)
