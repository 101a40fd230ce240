"""CPU ORACLE for the training step — test infrastructure, NOT product code (only tests/ may import it).

Restates ``FastSpeech2.training_step`` (litfass/fastspeech2/fastspeech2.py:786-797: result = self(batch);
losses = self.loss(result, batch); return losses["total"]) + Lightning's backward, ``clip_grad_norm_`` (gradient_clip_val,
scripts/train.sh:16) and ``configure_optimizers`` (fastspeech2.py:1166-1182: AdamW(betas=[0.9, 0.98], eps=1e-8,
weight_decay=0.01) + NoamLR, noam.py:19-25) on top of the forward oracle (oracle_cpu.forward with teacher targets) with
torch autograd doing the differentiation, dropout off.

Pinning: tools/gen_golden_train.py runs the REAL reference (its FastSpeech2.forward(inference=False), its FastSpeech2Loss,
loss.backward(), torch.nn.utils.clip_grad_norm_, torch.optim.AdamW and its NoamLR) in the build container and commits the
per-parameter gradients, the losses and the weights after three optimizer steps as tests/golden/train_small.npz;
tests/test_train_oracle.py checks this file against it.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import oracle_cpu

DEFAULT_ALPHAS = {"mel": 1.0, "pitch": 1e-1, "energy": 1e-1, "snr": 1e-1, "duration": 1e-4}


class _SoftDtwChunk(torch.autograd.Function):
    """One (prediction chunk, target chunk) batch through oracle.softdtw_cpu: value forward, d value / d prediction backward
    (= the reference's vendored _SoftDTW + calc_distance_matrix, pinned on tests/golden/softdtw_grad_small.npz)."""

    @staticmethod
    def forward(ctx, x, y, gamma):
        from . import softdtw_cpu
        val, grad = softdtw_cpu.soft_dtw_value_and_grad(x.detach().numpy(), y.detach().numpy(), gamma)
        ctx.save_for_backward(torch.from_numpy(grad))
        return torch.from_numpy(val)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g.view(-1, 1, 1) * grad, None, None


def _soft_dtw(pred, truth, valid, gamma, chunk):
    """get_loss with loss == "soft_dtw" (loss.py:62-81): zero-filled pads, chunks of `chunk` frames, summed over chunks and batch."""
    if pred.dim() == 2:
        pred, truth = pred.unsqueeze(-1), truth.unsqueeze(-1)
    v = valid.unsqueeze(-1)
    pred, truth = pred.masked_fill(~v, 0), truth.masked_fill(~v, 0)
    total = None
    for pc, tc in zip(pred.split(chunk, dim=1), truth.split(chunk, dim=1)):
        val = _SoftDtwChunk.apply(pc.contiguous(), tc.contiguous(), gamma)
        total = val if total is None else total + val
    return total.sum()


def _masked(pred, truth, kind, valid, soft_dtw_gamma=0.01, soft_dtw_chunk_size=256):
    """get_loss, loss.py:57-81: masked_select by the valid mask, then nn.L1Loss / nn.MSELoss (mean)."""
    if kind == "soft_dtw":
        return _soft_dtw(pred, truth, valid, soft_dtw_gamma, soft_dtw_chunk_size)
    if pred.dim() == 3:
        valid = valid.unsqueeze(-1).expand_as(pred)
    d = pred[valid] - truth[valid]
    return d.abs().mean() if kind == "l1" else (d * d).mean()


def losses(cfg, result, batch, variance_losses=None, mel_loss="l1", duration_loss="mse", alphas=None, soft_dtw_gamma=0.01,
           soft_dtw_chunk_size=256, loss_alphas=None):
    """``loss_alphas`` is the reference's keyword for ``alphas`` (loss.py:17-26; FastSpeech2.__init__ fills it from
    mel_loss_weight / duration_loss_weight / variance_loss_weights, fastspeech2.py:445-451)."""
    alphas = dict(DEFAULT_ALPHAS if (alphas is None and loss_alphas is None) else (alphas if alphas is not None else loss_alphas))
    variance_losses = variance_losses or ["mse"] * len(cfg.variances)
    tgt_valid, src_valid = ~result["tgt_mask"], ~result["src_mask"]
    out = OrderedDict()
    for v, kind in zip(cfg.variances, variance_losses):
        out[v] = _masked(result[f"variances_{v}"], torch.as_tensor(np.asarray(batch[f"variances_{v}"])).float(), kind, tgt_valid,
                         soft_dtw_gamma, soft_dtw_chunk_size)
    out["mel"] = _masked(result["mel"], torch.as_tensor(np.asarray(batch["mel"])).float(), mel_loss, tgt_valid, soft_dtw_gamma,
                         soft_dtw_chunk_size)
    dur_t = torch.log(torch.as_tensor(np.asarray(batch["duration"])).float() + 1)
    out["duration"] = _masked(result["duration_prediction"], dur_t, duration_loss, src_valid, soft_dtw_gamma, soft_dtw_chunk_size)
    out["total"] = sum(v * alphas[k] for k, v in out.items())
    return out


class OracleTrainer:
    def __init__(self, cfg, state_dict, lr=2e-4, warmup_steps=4000, gradient_clip_val=1.0, dtype=torch.float32, **loss_kw):
        self.cfg, self.loss_kw, self.clip, self.warmup, self.base_lr = cfg, loss_kw, gradient_clip_val, warmup_steps, lr
        self.sd = OrderedDict()
        for k, v in state_dict.items():
            t = torch.as_tensor(np.asarray(v)).clone()
            if t.is_floating_point():
                t = t.to(dtype)
                if not (k.endswith(".pe") or k.endswith(".bins")):
                    t.requires_grad_(True)
            self.sd[k] = t
        self.params = [t for t in self.sd.values() if t.requires_grad]
        self.opt = torch.optim.AdamW(self.params, lr=lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
        self.steps = 0

    def training_step(self, batch):
        tt = {k: batch[k] for k in batch if k == "duration" or k.startswith("variances_")}
        pri = {k: batch[k] for k in batch if k.startswith("priors_")}
        res = oracle_cpu.forward(self.sd, self.cfg, batch["phones"], batch["speaker"], teacher_targets=tt, priors=pri or None)
        ls = losses(self.cfg, res, batch, **self.loss_kw)
        ls["total"].backward()
        return {k: float(v.detach()) for k, v in ls.items()}, res

    def gradients(self):
        return OrderedDict((k, (t.grad if t.grad is not None else torch.zeros_like(t)).detach().clone())
                           for k, t in self.sd.items() if t.requires_grad)

    def optimizer_step(self, accum=1):
        e = max(1, self.steps)
        lr = self.base_lr * self.warmup ** 0.5 * min(e ** -0.5, e * self.warmup ** -1.5)
        for g in self.opt.param_groups:
            g["lr"] = lr
        if accum != 1:
            for p_ in self.params:
                if p_.grad is not None:
                    p_.grad /= accum
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.steps += 1
        return lr
