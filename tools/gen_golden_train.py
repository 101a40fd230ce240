#!/usr/bin/env python
"""Golden vectors for the training step: runs the REFERENCE itself (imported from /root/reference, build
container only) - its FastSpeech2.forward(batch) in teacher-forced mode, its FastSpeech2Loss, loss.backward(),
torch.nn.utils.clip_grad_norm_(1.0) (scripts/train.sh:16), torch.optim.AdamW as configure_optimizers builds it
(fastspeech2.py:1166-1173) and its own NoamLR (noam.py) - on the batch of tests/golden/teacher_small.npz plus a seeded mel
target, dropout off (eval mode: the reference's dropouts are random per step and cannot be pinned).

    python tools/gen_golden_train.py      ->  tests/golden/train_small.npz

The fixture holds the batch, the losses and per-parameter gradients of step 1, and every parameter after three optimizer
steps (lr 2e-3, warmup 4: the third step runs at a different Noam rate; every step clips, the gradient norms are stored)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightningfastspeech2_amd.config import Fs2Config  # noqa: E402
from lightningfastspeech2_amd.weights import synth_state_dict  # noqa: E402
from tools import ref_import  # noqa: E402

LR, WARMUP, CLIP = 2e-3, 4, 1.0


def run_case(name, cfg, sd, synth_json, batch, FastSpeech2Loss, NoamLR, loss_alphas=None):
    model = ref_import.build_reference_model(cfg, sd)  # eval mode: dropout off
    nv = len(cfg.variances)
    kw = {} if loss_alphas is None else {"loss_alphas": dict(loss_alphas)}  # what FastSpeech2.__init__ builds, fastspeech2.py:445-451
    loss = FastSpeech2Loss(variances=list(cfg.variances), variance_levels=["frame"] * nv, variance_transforms=["none"] * nv,
                           variance_losses=["mse"] * nv, mel_loss="l1", duration_loss="mse", max_length=4096, **kw)
    opt = torch.optim.AdamW(model.parameters(), lr=LR, betas=[0.9, 0.98], eps=1e-8, weight_decay=0.01)
    sched = NoamLR(opt, WARMUP)
    out = {"config_json": np.array(cfg.to_json()), "synth_json": np.array(synth_json), "hyper_json": np.array(json.dumps(
        dict(lr=LR, warmup_steps=WARMUP, gradient_clip_val=CLIP, **({} if loss_alphas is None else {"loss_alphas": {
            k: v for k, v in loss_alphas.items() if k != "speakers"}}))))}
    for k, v in batch.items():
        out["in_" + k] = v.numpy()
    for step in (1, 2, 3):
        np.random.seed(0)
        result = model(batch)
        losses = loss(result, batch)
        opt.zero_grad()
        losses["total"].backward()
        if step == 1:
            for k, v in losses.items():
                out[f"loss_{k}"] = np.float64(v.item())
            for n, p in model.named_parameters():
                if n.startswith("fastdiff_linear") or not p.requires_grad:
                    continue
                out["grad_" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP)
        out[f"gradnorm_{step}"] = np.float64(float(norm))
        out[f"lr_{step}"] = np.float64(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        print(f"{name} step {step}: total={float(losses['total'].detach()):.6f} grad norm={float(norm):.4f} lr={out[f'lr_{step}']:.3e}")
    for n, p in model.named_parameters():
        if not n.startswith("fastdiff_linear") and p.requires_grad:
            out["after3_" + n] = p.detach().numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(path, f"{os.path.getsize(path) / 1024:.1f} KiB")


def main():
    assert ref_import.reference_available(), "needs /root/reference (build container only)"
    ref_import._install_stubs()
    sys.modules["pysdtw"].SoftDTW = lambda *a, **k: None
    from litfass.fastspeech2.loss import FastSpeech2Loss
    from litfass.fastspeech2.noam import NoamLR

    only = sys.argv[1:]
    if not only or "train_small" in only:
        _dense(FastSpeech2Loss, NoamLR)
    _rest(FastSpeech2Loss, NoamLR, only)


def _recipe(FastSpeech2Loss, NoamLR, only):
    """The shipped recipe's architecture (scripts/train.sh:12-13,27-36,49: four variances incl. srmr, five priors, five-layer
    duration predictor, six decoder layers, depth-wise blocks) on the batch of recipe_teacher_small.npz."""
    if only and "train_recipe_small" not in only:
        return
    z = np.load(os.path.join(ROOT, "tests", "golden", "recipe_teacher_small.npz"))
    cfg = Fs2Config.from_json(str(z["config_json"]))
    skw = json.loads(str(z["synth_json"]))
    sd = synth_state_dict(cfg, skw.pop("seed"), **skw)
    B, T = z["out_mel"].shape[:2]
    rs = np.random.RandomState(778)
    batch = {"phones": torch.from_numpy(z["phones"]), "speaker": torch.from_numpy(z["speaker"]),
             "duration": torch.from_numpy(z["tf_duration"]),
             "mel": torch.from_numpy((rs.randn(B, T, cfg.n_mels) * 1.3 - 2.0).astype(np.float32))}
    for v in cfg.variances:
        batch[f"variances_{v}"] = torch.from_numpy(z[f"tf_variances_{v}"])
    for k in z.files:
        if k.startswith("in_priors_"):
            batch[k[3:]] = torch.from_numpy(z[k])
    # scripts/train.sh:25-26: --variance_loss_weights 1 1 1 1 --duration_loss_weight 1; mel_loss_weight's default is 1
    alphas = {"mel": 1.0, "duration": 1.0, "speakers": 1.0, **{v: 1.0 for v in cfg.variances}}
    run_case("train_recipe_small", cfg, sd, str(z["synth_json"]), batch, FastSpeech2Loss, NoamLR, loss_alphas=alphas)


def _dense(FastSpeech2Loss, NoamLR):
    # dense family: the batch of teacher_small.npz
    z = np.load(os.path.join(ROOT, "tests", "golden", "teacher_small.npz"))
    cfg = Fs2Config.from_json(str(z["config_json"]))
    skw = json.loads(str(z["synth_json"]))
    sd = synth_state_dict(cfg, skw.pop("seed"), **skw)
    B, T = z["out_mel"].shape[:2]
    rs = np.random.RandomState(777)
    batch = {"phones": torch.from_numpy(z["phones"]), "speaker": torch.from_numpy(z["speaker"]),
             "duration": torch.from_numpy(z["tf_duration"]),
             "mel": torch.from_numpy((rs.randn(B, T, cfg.n_mels) * 1.3 - 2.0).astype(np.float32))}
    for v in cfg.variances:
        batch[f"variances_{v}"] = torch.from_numpy(z[f"tf_variances_{v}"])
    run_case("train_small", cfg, sd, str(z["synth_json"]), batch, FastSpeech2Loss, NoamLR)


def _rest(FastSpeech2Loss, NoamLR, only):
    _recipe(FastSpeech2Loss, NoamLR, only)
    if only and "train_dw_small" not in only:
        return
    # depth-wise family (the reference's class defaults, fastspeech2.py:68,76,98,107): every conv depth-wise, odd kernel mix
    from lightningfastspeech2_amd.weights import synth_inputs
    cfg = Fs2Config(n_phones=40, encoder_hidden=64, decoder_hidden=64, encoder_head=2, decoder_head=2, encoder_layers=2,
                    decoder_layers=2, encoder_kernel_sizes=[5, 9], decoder_kernel_sizes=[7, 3], encoder_conv_filter_size=256,
                    decoder_conv_filter_size=128, encoder_depthwise_conv=True, decoder_depthwise_conv=True,
                    variance_filter_size=64, variance_depthwise_conv=True, variance_nlayers=[2, 2, 2], variance_kernel_size=[3, 5, 3],
                    duration_filter_size=64, duration_depthwise_conv=True, duration_nlayers=2, variance_nbins=16, n_mels=8,
                    stats={"pitch": {"min": -2.0, "max": 2.5, "mean": 0.1, "std": 1.5},
                           "energy": {"min": -3.0, "max": 3.0, "mean": 0.0, "std": 1.0},
                           "snr": {"min": -1.0, "max": 4.0, "mean": 1.2, "std": 2.0}})
    skw = dict(seed=11, randomize_norm=True, duration_bias=1.0)
    sd = synth_state_dict(cfg, 11, randomize_norm=True, duration_bias=1.0)
    B, L, lengths = 3, 13, [13, 9, 5]
    inp = synth_inputs(cfg, B, L, seed=2011, lengths=lengths)
    rs = np.random.RandomState(8011)
    dur = rs.randint(0, 6, size=(B, L)).astype(np.int64)
    for b, n in enumerate(lengths):
        dur[b, n:] = 0
    T = int(dur.sum(1).max())
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"]), "duration": torch.from_numpy(dur),
             "mel": torch.from_numpy((rs.randn(B, T, cfg.n_mels) * 1.3 - 2.0).astype(np.float32))}
    for v in cfg.variances:
        batch[f"variances_{v}"] = torch.from_numpy((1.2 * rs.randn(B, T)).astype(np.float32))
    run_case("train_dw_small", cfg, sd, json.dumps(skw, sort_keys=True), batch, FastSpeech2Loss, NoamLR)


if __name__ == "__main__":
    main()
