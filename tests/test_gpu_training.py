"""GPU: the HIP training step (lightningfastspeech2_amd/training.py over the backward operators of include/fs2.h) against
(a) the fixture the REAL reference produced (tests/golden/train_small.npz: losses, every parameter's gradient, the weights
after three clipped AdamW + Noam steps) and (b) the CPU oracle (oracle/train_cpu.py) on other shapes: ragged lengths, odd
kernel sizes, gradient accumulation.  fp32 arithmetic; tolerances: gradients 1e-4 of the tensor's largest entry
(the chain is ~40 fp32 GEMMs deep), losses 1e-5 relative, weights 2e-5 where Adam is well conditioned."""
import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.config import Fs2Config
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
from oracle import train_cpu
from test_train_oracle import CASES, assert_params_close, load

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-4


def _dev(batch):
    return {k: torch.as_tensor(v).cuda() for k, v in batch.items()}


def _check_grads(got, want, tol=GRAD_TOL):
    assert sorted(got) == sorted(want)
    worst = ("", 0.0)
    for n, w in want.items():
        w = torch.as_tensor(w).float()
        err = float((got[n] - w).abs().max()) / (float(w.abs().max()) + 1e-3)
        if err > worst[1]:
            worst = (n, err)
    assert worst[1] <= tol, worst


@pytest.mark.parametrize("name", CASES)
def test_training_step_matches_reference_fixture(name):
    from lightningfastspeech2_amd.training import Trainer
    z, cfg, sd, batch, hyper = load(name)
    tr = Trainer(cfg, sd, **hyper)
    for step in (1, 2, 3):
        losses = tr.training_step(_dev(batch))
        if step == 1:
            for k, v in losses.items():
                w = float(z[f"loss_{k}"])
                assert abs(float(v) - w) <= 1e-5 * max(1.0, abs(w)), (k, float(v), w)
            _check_grads(tr.gradients(), {k[5:]: z[k] for k in z.files if k.startswith("grad_")})
        norm = float(torch.sqrt((tr.flat_g.double() ** 2).sum()))
        assert abs(norm - float(z[f"gradnorm_{step}"])) <= 2e-4 * float(z[f"gradnorm_{step}"])
        lr = tr.optimizer_step()
        assert abs(lr - float(z[f"lr_{step}"])) <= 1e-12
    after = tr.state_dict()
    for k in z.files:
        if k.startswith("after3_"):
            assert_params_close(z, k[7:], after[k[7:]], 2e-5)


def _case(seed, B, L, lengths, **kw):
    base = dict(n_phones=30, encoder_hidden=64, decoder_hidden=64, encoder_head=2, decoder_head=4, encoder_layers=1,
                decoder_layers=2, encoder_kernel_sizes=[5], decoder_kernel_sizes=[9, 3], encoder_conv_filter_size=96,
                decoder_conv_filter_size=160, encoder_depthwise_conv=False, decoder_depthwise_conv=False,
                variance_filter_size=64, variance_depthwise_conv=False, variance_nlayers=[2, 1], variances=["pitch", "energy"],
                variance_levels=["frame", "frame"], variance_transforms=["none", "none"], variance_kernel_size=[3, 5],
                duration_filter_size=64, duration_depthwise_conv=False, duration_nlayers=2, variance_nbins=24, n_mels=20,
                stats={"pitch": {"min": -2.0, "max": 2.5, "mean": 0.1, "std": 1.5},
                       "energy": {"min": -3.0, "max": 3.0, "mean": 0.0, "std": 1.0}})
    base.update(kw)
    cfg = Fs2Config(**base)
    sd = synth_state_dict(cfg, seed, randomize_norm=True, duration_bias=1.0)
    inp = synth_inputs(cfg, B, L, seed=seed + 1, lengths=lengths)
    rs = np.random.RandomState(seed + 2)
    dur = rs.randint(0, 5, size=(B, L)).astype(np.int64)
    for b, n in enumerate(lengths):
        dur[b, n:] = 0
    dur[0, 0] = max(1, dur[0, 0])
    T = int(dur.sum(1).max())
    batch = {"phones": inp["phones"], "speaker": inp["speaker"], "duration": dur,
             "mel": (rs.randn(B, T, cfg.n_mels) - 1.5).astype(np.float32)}
    batch.update({k: v for k, v in inp.items() if k.startswith("priors_")})
    for v in cfg.variances:
        batch[f"variances_{v}"] = (1.1 * rs.randn(B, T)).astype(np.float32)
    return cfg, sd, batch


PRIORS = dict(priors=["pitch", "duration"],
              stats={"pitch": {"min": -2.0, "max": 2.5, "mean": 0.1, "std": 1.5}, "energy": {"min": -3.0, "max": 3.0, "mean": 0.0, "std": 1.0},
                     "pitch_prior": {"min": -1.0, "max": 1.0}, "duration_prior": {"min": 0.0, "max": 5.0}})
DW = dict(encoder_depthwise_conv=True, decoder_depthwise_conv=True, variance_depthwise_conv=True, duration_depthwise_conv=True,
          encoder_conv_filter_size=128, decoder_conv_filter_size=192, decoder_kernel_sizes=[17, 3])


@pytest.mark.parametrize("seed,B,L,lengths,kw", [(3, 4, 13, [13, 9, 5, 1], {}), (8, 2, 37, [37, 20], {}), (5, 3, 21, [21, 8, 2], DW),
                                                 (6, 2, 9, [9, 4], dict(DW, decoder_depthwise_conv=False, variance_depthwise_conv=False)),
                                                 (7, 3, 10, [10, 6, 8], PRIORS)])
def test_training_step_matches_oracle_on_ragged_batches(seed, B, L, lengths, kw):
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, batch = _case(seed, B, L, lengths, **kw)
    kw = dict(lr=1e-3, warmup_steps=2, gradient_clip_val=0.5, variance_losses=["l1", "mse"], mel_loss="mse", duration_loss="l1")
    ref = train_cpu.OracleTrainer(cfg, sd, **kw)
    want_l, _ = ref.training_step(batch)
    tr = Trainer(cfg, sd, **kw)
    got_l = tr.training_step(_dev(batch))
    for k, w in want_l.items():
        assert abs(float(got_l[k]) - w) <= 1e-5 * max(1.0, abs(w)), (k, float(got_l[k]), w)
    _check_grads(tr.gradients(), ref.gradients())


def test_gradient_accumulation_and_bit_equal_reruns():
    """Two micro-batches accumulate as Lightning's accumulate_grad_batches does (mean of the two gradients feeds the clip
    and AdamW); the same step twice from the same state gives bit-equal gradients (fixed-order reductions everywhere)."""
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, b1 = _case(21, 3, 11, [11, 6, 2])
    _, _, b2 = _case(22, 3, 11, [11, 10, 7])
    kw = dict(lr=1e-3, warmup_steps=2, gradient_clip_val=1.0)
    ref = train_cpu.OracleTrainer(cfg, sd, **kw)
    ref.training_step(b1)
    ref.training_step(b2)
    want = ref.gradients()
    tr = Trainer(cfg, sd, **kw)
    tr.training_step(_dev(b1))
    g1 = tr.flat_g.clone()
    tr.training_step(_dev(b2))
    _check_grads(tr.gradients(), want)
    ref.optimizer_step(accum=2)
    tr.optimizer_step()
    after = tr.state_dict()
    for n, t in ref.sd.items():
        if t.requires_grad:
            g = want[n].abs()
            solid = g > 1e-5 * max(1.0, float(g.max()))
            d = (after[n] - t.detach().float()).abs()
            assert float(d[solid].max()) <= 2e-5, n
    tr2 = Trainer(cfg, sd, **kw)
    tr2.training_step(_dev(b1))
    assert torch.equal(tr2.flat_g, g1)


def test_trainer_rejects_what_is_not_built():
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, _ = _case(1, 2, 5, [5, 3])
    with pytest.raises(NotImplementedError):
        Trainer(cfg, sd, mel_loss="huber")
    with pytest.raises(ValueError, match="even kernel size"):   # 'same' padding of an even kernel is asymmetric
        import dataclasses
        Trainer(dataclasses.replace(cfg, decoder_kernel_sizes=[4, 3]), sd)
    with pytest.raises(ValueError, match="loss_alphas"):
        Trainer(cfg, sd, loss_alphas={"mel": 1.0, "duration": 1.0})
    assert Trainer(cfg, sd, gradient_clip_val=0).gradient_clip_val is None   # Lightning: 0 = no clipping


def test_bf16_mixed_precision_step_tracks_the_fp32_gradients():
    """precision="bf16": bf16 activations / activation gradients / GEMM operands, fp32 masters, weight gradients and losses.
    Not a 1e-4 mode: per parameter tensor the gradient must point the same way as the oracle's (cosine >= 0.99 wherever the
    gradient is not numerical noise) and the losses agree to 2 %; three optimizer steps then reduce the loss."""
    from lightningfastspeech2_amd.training import Trainer
    z, cfg, sd, batch, hyper = load()
    ref = train_cpu.OracleTrainer(cfg, sd, **hyper)
    want_l, _ = ref.training_step(batch)
    want = ref.gradients()
    tr = Trainer(cfg, sd, precision="bf16", **hyper)
    got_l = tr.training_step(_dev(batch))
    for k, w in want_l.items():
        assert abs(float(got_l[k]) - w) <= 2e-2 * max(1.0, abs(w)), (k, float(got_l[k]), w)
    got = tr.gradients()
    worst = ("", 1.0)
    gmax = max(float(w.abs().max()) for w in want.values())
    for n, w in want.items():
        if float(w.abs().max()) < 1e-4 * gmax:
            continue  # numerically-zero gradients (key biases): direction is noise
        cos = float((got[n].double() * w.double()).sum() / (got[n].double().norm() * w.double().norm() + 1e-30))
        if cos < worst[1]:
            worst = (n, cos)
    assert worst[1] >= 0.99, worst
    first = float(got_l["total"])
    tr.optimizer_step()
    for _ in range(3):
        last = float(tr.training_step(_dev(batch))["total"])
        tr.optimizer_step()
    assert last < first


def test_dropout_op_statistics_and_determinism():
    import ctypes as C
    from lightningfastspeech2_amd import _lib
    lib = _lib.load()
    n, p = 1 << 20, 0.3
    x = torch.ones(n, device="cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for seed, key in ((5, 1), (5, 1), (5, 2), (6, 1)):
        y = torch.empty_like(x)
        _lib.check(lib.fs2_op_dropout(_lib.FS2_F32, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, C.c_float(p), C.c_uint64(seed), C.c_uint64(key), st))
        outs.append(y.cpu())
    a, b, c, d = outs
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    keep = float((a != 0).float().mean())
    assert abs(keep - (1 - p)) <= 4 * (p * (1 - p) / n) ** 0.5
    assert torch.allclose(a[a != 0], torch.tensor(1 / (1 - p)))
    both = float(((a != 0) & (c != 0)).float().mean())   # two sites: independent masks
    assert abs(both - (1 - p) ** 2) <= 5e-3
    xb = torch.ones(n, device="cuda:0", dtype=torch.bfloat16)
    yb = torch.empty_like(xb)
    _lib.check(lib.fs2_op_dropout(_lib.FS2_BF16, C.c_void_p(xb.data_ptr()), C.c_void_p(yb.data_ptr()), n, C.c_float(p), C.c_uint64(5), C.c_uint64(1), st))
    assert torch.equal(yb.cpu() != 0, a != 0)  # the mask depends on (seed, key, index) only: an fp32 gradient meets the bf16 mask


@pytest.mark.parametrize("kw", [{}, DW])
def test_backward_is_the_derivative_of_the_forward_under_dropout(kw):
    """With every dropout site switched on (fixed seed, so the step is a deterministic function of the weights) the gradient
    buffer must be the directional derivative of the total loss: (L(w + eps v) - L(w - eps v)) / (2 eps) = g . v.  The
    directions are the gradient itself re-weighted element by element (v = g * u, u ~ U(0.5, 1.5)), so g . v = sum g^2 u is far
    above the fp32 noise of a loss difference and a wrong mask, scale or position in the chain at any site shows up in it;
    tools/probes/fd_check.py does the same with random directions, site by site."""
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, batch = _case(31, 3, 12, [12, 7, 3], **kw)
    drop = dict(encoder_dropout=0.2, decoder_dropout=0.15, variance_dropout=[0.3, 0.25], duration_dropout=0.3, seed=9)
    tr = Trainer(cfg, sd, gradient_clip_val=None, **drop)
    bd = _dev(batch)
    l0 = float(tr.training_step(bd)["total"])
    g = tr.flat_g.clone().double()
    tr.zero_grad()
    tr0 = Trainer(cfg, sd, gradient_clip_val=None)
    assert abs(float(tr0.training_step(bd)["total"]) - l0) > 1e-3  # dropout really is on
    w0 = tr.flat_p.clone()
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    for trial in range(3):
        v = g.float() * (0.5 + torch.rand(tr.n_flat, device="cuda:0", generator=gen))
        eps = 1e-4 / float(v.norm()) * float(w0.norm())  # |eps v| = 1e-4 |w|: few ReLU / L1 kinks are crossed
        vals = []
        for sgn in (1.0, -1.0):
            tr.flat_p.copy_(w0 + sgn * eps * v)
            tr._refresh_shadow()
            tr._micro = 0  # same masks as the step whose gradient is checked
            vals.append(float(tr.training_step(bd)["total"].double()))
            tr.zero_grad()
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g * v.double()).sum())
        assert an > 0 and abs(fd - an) <= 3e-2 * an, (trial, fd, an)
    tr.flat_p.copy_(w0)


@pytest.mark.parametrize("dw", [False, True])
def test_weight_gradient_stream_is_bit_identical(monkeypatch, dw):
    """r04: the weight-gradient GEMMs (+ split-K reduce, bias column sums, embedding scatters, depth-wise weight gradients) - leaves of
    the backward - run on a second HIP stream beside the data-gradient chain (FS2_TRAIN_WGRAD_STREAM, on for bf16).  Same kernels on
    the same inputs, ordered by events: losses, every gradient and the weights after two steps are bit-equal to the one-stream run,
    with dropout on (in-place mask passes over tensors the side stream still reads are guarded)."""
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, batch = _case(61, 3, 17, [17, 11, 4], encoder_depthwise_conv=dw, decoder_depthwise_conv=dw, variance_depthwise_conv=dw,
                           duration_depthwise_conv=dw, encoder_conv_filter_size=128, decoder_conv_filter_size=128)
    outs = []
    for on in ("0", "1"):
        monkeypatch.setenv("FS2_TRAIN_WGRAD_STREAM", on)
        tr = Trainer(cfg, sd, precision="bf16", encoder_dropout=0.1, decoder_dropout=0.1, variance_dropout=0.1, duration_dropout=0.1, seed=5)
        assert (tr.ops.side is not None) == (on == "1")
        l1 = tr.training_step(_dev(batch))
        g1 = {k: v.clone() for k, v in tr.gradients().items()}
        tr.optimizer_step()
        tr.training_step(_dev(batch))
        tr.optimizer_step()
        torch.cuda.synchronize()
        outs.append((l1, g1, tr.state_dict()))
    (la, ga, wa), (lb, gb, wb) = outs
    for k in la:
        assert float(la[k]) == float(lb[k]), k
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    for k in wa:
        assert torch.equal(torch.as_tensor(wa[k]), torch.as_tensor(wb[k])), k


def test_two_rank_data_parallel_step(tmp_path):
    """Two ranks (both on this box's one GPU, gloo), each with its shard of a ragged batch: per-rank step, the flat gradient
    buffer all-reduced inside optimizer_step.  The result must equal the oracle that accumulates the two shards' gradients
    and steps with their mean (DDP averages per-rank mean losses) - and be identical on both ranks."""
    import os
    import socket
    import subprocess
    import sys
    from lightningfastspeech2_amd.dist import shard_batch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "after.npz"
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "_dist_train_worker.py"), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    cfg, sd, batch = _case(41, 4, 11, [11, 9, 6, 2])
    ref = train_cpu.OracleTrainer(cfg, sd, lr=1e-3, warmup_steps=2, gradient_clip_val=1.0)
    full = {k: torch.as_tensor(v) for k, v in batch.items()}
    frames = full["duration"].sum(1)  # zero pads, as the collate format has them (the worker does the same; shard_batch refuses to cut non-zero frames)
    for k in list(full):
        if k == "mel" or k.startswith("variances_"):
            keep = torch.arange(full[k].shape[1])[None, :] < frames[:, None]
            full[k] = full[k] * (keep[..., None] if full[k].dim() == 3 else keep)
    for rank in range(2):
        mine = shard_batch(full, 2, rank, trim=True)
        T = int(mine["duration"].sum(1).max())
        for k in list(mine):
            if k == "mel" or k.startswith("variances_"):
                mine[k] = mine[k][:, :T].contiguous()
        ref.training_step({k: v.numpy() for k, v in mine.items()})
    grads = ref.gradients()
    ref.optimizer_step(accum=2)
    for n, t in ref.sd.items():
        if t.requires_grad:
            g = grads[n].abs()
            solid = g > 1e-5 * max(1.0, float(g.max()))
            d = (torch.from_numpy(got[n]) - t.detach().float()).abs()
            assert float(d[solid].max()) <= 2e-5, n


@pytest.mark.parametrize("drop", [0.0, 0.15])
def test_flash_attention_path_matches_the_materialised_path(drop):
    """Head dim 128 in bf16 takes the fused attention forward + recomputing backward; the materialised path (probabilities in
    HBM, strided-batched GEMMs) stays as its cross-check: same masks (same seed), so per tensor the two gradients must agree
    closely (cosine >= 0.999 on the attention projections and in the median, >= 0.99 everywhere), and both track the fp32 oracle
    without dropout."""
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, batch = _case(51, 3, 21, [21, 13, 5], encoder_hidden=256, decoder_hidden=256, variance_filter_size=256, duration_filter_size=256,
                           encoder_conv_filter_size=256, decoder_conv_filter_size=256, decoder_head=2)
    kw = dict(precision="bf16", gradient_clip_val=None, encoder_dropout=drop, decoder_dropout=drop, seed=4)
    a, b = Trainer(cfg, sd, **kw), Trainer(cfg, sd, attention="materialized", **kw)
    la, lb = a.training_step(_dev(batch)), b.training_step(_dev(batch))
    assert abs(float(la["total"]) - float(lb["total"])) <= 5e-3 * abs(float(lb["total"]))
    ga, gb = a.gradients(), b.gradients()
    gmax = max(float(v.abs().max()) for v in gb.values())
    worst, coses = ("", 1.0), []
    for n in gb:
        if float(gb[n].abs().max()) < 1e-4 * gmax:
            continue
        cos = float((ga[n].double() * gb[n].double()).sum() / (ga[n].double().norm() * gb[n].double().norm() + 1e-30))
        coses.append(cos)
        if cos < worst[1]:
            worst = (n, cos)
    assert worst[1] >= 0.99, worst               # cancellation-prone sums (predictor biases) wobble with any bf16 change upstream
    assert float(np.median(coses)) >= 0.9995, float(np.median(coses))
    for n in gb:                                  # the attention projections themselves
        if "in_proj_weight" in n or "out_proj.weight" in n:
            cos = float((ga[n].double() * gb[n].double()).sum() / (ga[n].double().norm() * gb[n].double().norm() + 1e-30))
            assert cos >= 0.999, (n, cos)
    if drop == 0.0:
        ref = train_cpu.OracleTrainer(cfg, sd, gradient_clip_val=None)
        ref.training_step(batch)
        want = ref.gradients()
        for n, w in want.items():
            if float(w.abs().max()) < 1e-4 * gmax:
                continue
            cos = float((ga[n].double() * w.double()).sum() / (ga[n].double().norm() * w.double().norm() + 1e-30))
            assert cos >= 0.99, (n, cos)


def test_checkpoint_resume_is_exact():
    """state_dict() + optimizer_state() of a run after two steps, loaded into a fresh Trainer: the third step (with dropout on,
    so the mask counter matters too) leaves bit-identical weights in both."""
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, batch = _case(61, 3, 12, [12, 8, 3])
    kw = dict(lr=1e-3, warmup_steps=2, encoder_dropout=0.1, decoder_dropout=0.1, variance_dropout=0.2, duration_dropout=0.2, seed=5)
    a = Trainer(cfg, sd, **kw)
    bd = _dev(batch)
    for _ in range(2):
        a.training_step(bd)
        a.optimizer_step()
    b = Trainer(cfg, a.state_dict(), **kw)
    b.load_optimizer_state(a.optimizer_state())
    for t in (a, b):
        t.training_step(bd)
        t.optimizer_step()
    assert torch.equal(a.flat_p, b.flat_p) and torch.equal(a.flat_m, b.flat_m) and a.steps == b.steps == 3


@pytest.mark.parametrize("case", ["one_full", "ragged3"])
def test_full_length_utterance_gradients_vs_oracle(case):
    """BASELINE configs[1] architecture (FS2-27M: H = 256, F = 1024, k = 9, 4 + 4 layers) on one full-length utterance
    (256 phonemes -> 1536 frames), and on a ragged batch of three (256 / 180 / 64 phonemes, durations 2..10 frames: pad phones, pad
    frames and unequal T at the full length), fp32: losses and every parameter's gradient against the autograd oracle.  At this length a
    gradient entry is a sum over 1536 frames of cancelling fp32 terms on BOTH sides (the oracle is fp32 too): per tensor the
    direction must agree to cosine >= 0.99999 and no entry may differ by more than 2 % of the tensor's largest (measured:
    cosine >= 0.999997, worst entry 1.0 %; tools/probes/full_len_grad_check.py prints the table)."""
    import math
    from lightningfastspeech2_amd.config import preset
    from lightningfastspeech2_amd.training import Trainer
    cfg = preset("c2")
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
    rs = np.random.RandomState(5)
    if case == "one_full":
        B = 1
        inp = synth_inputs(cfg, 1, 256, seed=1234)
        dur = np.full((1, 256), 6, np.int64)
    else:
        B, lengths = 3, [256, 180, 64]
        inp = synth_inputs(cfg, B, 256, seed=1234, lengths=lengths)
        dur = rs.randint(2, 11, size=(B, 256)).astype(np.int64)
        for b, n in enumerate(lengths):
            dur[b, n:] = 0
    T = int(dur.sum(1).max())
    batch = {"phones": inp["phones"], "speaker": inp["speaker"], "duration": dur,
             "mel": (rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)}
    for v in cfg.variances:
        batch[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
    ref = train_cpu.OracleTrainer(cfg, sd, gradient_clip_val=None)
    want_l, _ = ref.training_step(batch)
    tr = Trainer(cfg, sd, gradient_clip_val=None)
    got_l = tr.training_step(_dev(batch))
    for k, w in want_l.items():
        assert abs(float(got_l[k]) - w) <= 2e-5 * max(1.0, abs(w)), (k, float(got_l[k]), w)
    got, want = tr.gradients(), ref.gradients()
    gmax = max(float(w.abs().max()) for w in want.values())
    for n, w in want.items():
        w = w.float()
        if float(w.abs().max()) < 1e-5 * gmax:
            continue
        cos = float((got[n].double() * w.double()).sum() / (got[n].double().norm() * w.double().norm() + 1e-30))
        rel = float((got[n] - w).abs().max()) / float(w.abs().max())
        assert cos >= 0.99999 and rel <= 2e-2, (n, cos, rel)


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_baseline_configs_train_in_bf16(name):
    """BASELINE configs[2] (LS-76M, depth-wise, 6 heads) and configs[4] (FS2-1B, 12 + 12 layers) at their full architecture,
    a small batch: the bf16 step (fused attention path, folded conv2, every kernel-size bucket) gives finite losses, bit-equal
    gradients on a rerun from the same state, and three optimizer steps lower the loss."""
    import math
    from lightningfastspeech2_amd.config import preset
    from lightningfastspeech2_amd.training import Trainer
    cfg = preset(name)
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(4.0), duration_weight_scale=0.0)
    B, L, f = 2, 48, 3
    inp = synth_inputs(cfg, B, L, seed=77, lengths=[L, L - 11])
    rs = np.random.RandomState(3)
    dur = np.full((B, L), f, np.int64)
    dur[1, L - 11:] = 0
    T = L * f
    batch = {"phones": inp["phones"], "speaker": inp["speaker"], "duration": dur, "mel": (rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)}
    for v in cfg.variances:
        batch[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
    bd = _dev(batch)
    # Adam's first steps move every entry by ~lr whatever the gradient scale: at H = 1536 a 1e-3 step overshoots (the fp32 path and
    # the oracle's dynamics do the same, tools/probes/c5_loss_probe.py), so the 1B preset is stepped with a smaller rate
    tr = Trainer(cfg, sd, precision="bf16", lr=1e-3 if name == "c3" else 2e-5, warmup_steps=1)
    l0 = tr.training_step(bd)
    assert all(math.isfinite(float(v)) for v in l0.values())
    g0 = tr.flat_g.clone()
    assert bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0
    tr.zero_grad()
    tr._micro = 0
    tr.training_step(bd)
    assert torch.equal(tr.flat_g, g0)
    for _ in range(4):
        tr.optimizer_step()
        last = tr.training_step(bd)
    assert float(last["total"]) < float(l0["total"])


@pytest.mark.parametrize("name,dropout", [("c2", 0.0), ("c2", 0.1), ("ref-default", 0.0), ("ref-default", 0.1)])
def test_bf16_step_tracks_the_fp32_step_at_a_baseline_size(name, dropout):
    """bf16 training parity beyond the tiny fixture.  Full BASELINE configs[1] architecture (FS2-27M dense)
    and the reference-default depth-wise family, B = 4 utterances x 256 phonemes x 3 frames, ragged: the bf16 step against the
    fp32 HIP step from the same weights (the fp32 step is what the reference-produced fixtures pin to 1e-4) - every loss term
    within 2 % (measured: 1e-4 .. 1e-3), every parameter tensor's gradient cosine >= 0.98 and >= 0.99 for at least 90 % of the
    tensors (tensors whose gradient is numerical noise excepted), with the recipe's dropout too (same counter-based masks in
    both precisions).  Measured r03, C2 without dropout: 163 of 172 tensors >= 0.99; the nine below are the first conv layers of
    the variance predictors and the embedding tables they feed back into (0.986 .. 0.99), the same with materialised attention
    and with either data-gradient path - a property of the bf16 predictor chain, not of one kernel."""
    import math
    from lightningfastspeech2_amd.config import preset
    from lightningfastspeech2_amd.training import Trainer
    cfg = preset(name)
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(4.0), duration_weight_scale=0.0)
    B, L, f = 4, 256, 3
    lens = [L, L - 37, L - 90, L // 2]
    inp = synth_inputs(cfg, B, L, seed=91, lengths=lens)
    rs = np.random.RandomState(5)
    dur = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        dur[b, :n] = rs.randint(1, 2 * f, size=n)
    T = int(dur.sum(axis=1).max())
    batch = {"phones": inp["phones"], "speaker": inp["speaker"], "duration": dur, "mel": (rs.randn(B, T, cfg.n_mels) - 2).astype(np.float32)}
    for v in cfg.variances:
        batch[f"variances_{v}"] = rs.randn(B, T).astype(np.float32)
    bd = _dev(batch)
    kw = dict(lr=1e-3, warmup_steps=1, seed=11, encoder_dropout=dropout, decoder_dropout=dropout, duration_dropout=dropout,
              variance_dropout=dropout)
    ref = Trainer(cfg, sd, precision="fp32", **kw)
    want_l = {k: float(v) for k, v in ref.training_step(bd).items()}
    want = {n: g.double().cpu() for n, g in ref.gradients().items()}
    del ref
    torch.cuda.empty_cache()
    tr = Trainer(cfg, sd, precision="bf16", **kw)
    got_l = {k: float(v) for k, v in tr.training_step(bd).items()}
    for k, w in want_l.items():
        assert abs(got_l[k] - w) <= 2e-2 * max(1.0, abs(w)), (k, got_l[k], w)
    got = tr.gradients()
    gmax = max(float(w.abs().max()) for w in want.values())
    worst = ("", 1.0)
    checked = good = 0
    for n, w in want.items():
        if float(w.abs().max()) < 1e-4 * gmax:
            continue  # numerically-zero gradients (attention key biases): direction is noise
        g = got[n].double().cpu()
        cos = float((g * w).sum() / (g.norm() * w.norm() + 1e-30))
        checked += 1
        good += cos >= 0.99
        if cos < worst[1]:
            worst = (n, cos)
    # with the recipe's dropout the masks multiply every rounding difference by 1 / (1 - p) at ~40 sites: measured 149 of 172 >= 0.99
    assert checked >= 0.8 * len(want) and worst[1] >= 0.98 and good >= (0.9 if dropout == 0.0 else 0.8) * checked, (worst, good, checked, len(want))


def test_soft_dtw_loss_kind_trains():
    """mel_loss = "soft_dtw" and a soft-DTW variance loss (loss.py:36,62-81): zero-filled pads, chunks of soft_dtw_chunk_size
    frames, value summed over chunks and batch, the gradient through fs2_op_soft_dtw_grad - losses and every parameter's gradient
    against the oracle (torch autograd over the forward oracle with oracle.softdtw_cpu's value / gradient, itself pinned on the
    reference's vendored module); a chunk size below T exercises the chunk seam."""
    from lightningfastspeech2_amd.training import Trainer
    cfg, sd, batch = _case(2, 2, 9, [9, 6])
    kw = dict(mel_loss="soft_dtw", variance_losses=["soft_dtw", "mse"], soft_dtw_gamma=0.5, soft_dtw_chunk_size=16,
              lr=1e-3, warmup_steps=10)
    ref = train_cpu.OracleTrainer(cfg, sd, **kw)
    want_l, _ = ref.training_step(batch)
    want = ref.gradients()
    tr = Trainer(cfg, sd, **kw)
    got_l = tr.training_step(_dev(batch))
    for k, w in want_l.items():
        assert abs(float(got_l[k]) - w) <= 2e-4 * max(1.0, abs(w)), (k, float(got_l[k]), w)
    got = tr.gradients()
    for n, w in want.items():
        scale = float(w.abs().max()) + 1e-3 * max(float(v.abs().max()) for v in want.values())
        assert float((got[n].cpu() - w).abs().max()) <= 2e-3 * scale, n
    first = float(got_l["total"])
    tr.optimizer_step()
    for _ in range(3):
        last = float(tr.training_step(_dev(batch))["total"])
        tr.optimizer_step()
    assert last < first
