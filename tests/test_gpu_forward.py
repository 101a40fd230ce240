"""End-to-end parity (`-m gpu`): the HIP forward through the drop-in ``FastSpeech2`` object against
(a) the golden vectors captured from the real reference, (b) the CPU oracle on seeded inputs at
sizes it finishes in seconds, and (c) size-independent properties at BASELINE.json's full size.

Tolerance (BASELINE.json north_star): fp32 mode  max|mel - reference| <= 1e-3 over ALL (B,T,n_mels)
entries incl. pad rows, with duration_rounded / masks / bucket indices exactly equal.  bf16 mode is
the throughput mode: decisions may flip, so it is checked with the durations forced
to the oracle's and a bf16-sized tolerance.
"""
import json
import os

import numpy as np
import pytest
import torch

from _golden import Golden, golden_names, lookup
from lightningfastspeech2_amd.config import Fs2Config, preset
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
from oracle import oracle_cpu

pytestmark = pytest.mark.gpu

MEL_TOL_FP32 = 1e-3
# bf16 (throughput mode) under the oracle's decisions, measured r04 (gpurun_out/parity_report.jsonl): mel max-abs 0.014-0.018, mean
# 0.0023-0.0030 on O(1)-scale mels (scale 2.5-2.8), encoder output 0.054-0.062.  Asserted at <= 3.5x the measurement (VERDICT r04 weak 2:
# the bounds were 15-20x).
BF16_MEL_MAX, BF16_MEL_MEAN, BF16_ENC_MAX = 0.06, 0.008, 0.15
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def _report(**kw):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _model(cfg, sd, precision):
    from lightningfastspeech2_amd.model import FastSpeech2
    return FastSpeech2(cfg, sd, precision=precision, device="cuda:0")


def _cpu(d):
    return {k: ({kk: vv.cpu() for kk, vv in v.items()} if isinstance(v, dict) else v.cpu()) for k, v in d.items()}


@pytest.mark.parametrize("name", golden_names())
def test_fp32_matches_reference_golden(name):
    g = Golden(name)
    m = _model(g.cfg, g.state_dict(), "fp32")
    m.engine.set_debug(True)
    batch = {"phones": torch.from_numpy(g.phones), "speaker": torch.from_numpy(g.speaker), **g.priors}
    if g.teacher is not None:  # teacher-forced forward, as the Lightning hooks call it: model(batch)
        batch.update({k: torch.from_numpy(v) for k, v in g.teacher.items()})
        out = _cpu(m(batch))
    else:
        out = _cpu(m(batch, inference=True))
    errs = {}
    assert tuple(out["mel"].shape) == g.out["mel"].shape
    assert out["tgt_mask"].dtype == torch.bool
    assert out["duration_rounded"].dtype == (torch.int32 if g.teacher is None else torch.int64)
    for k in ("duration_rounded", "src_mask", "tgt_mask"):
        assert np.array_equal(out[k].numpy(), g.out[k]), k
    for k, ref in g.out.items():
        if ref.dtype.kind == "f":
            errs[k] = float(np.abs(lookup(out, k).numpy() - ref).max())
    for k, ref in g.mid.items():
        errs["mid_" + k] = float(np.abs(m.engine.debug_tensor(k).cpu().numpy() - ref).max())
    _report(test="golden_fp32", case=name, errs=errs)
    assert errs["mel"] <= MEL_TOL_FP32, errs
    for k, e in errs.items():
        assert e <= MEL_TOL_FP32, (k, e)


def _oracle_case(cfg, B, L, lengths, seed, **skw):
    sd = synth_state_dict(cfg, seed, randomize_norm=True, **skw)
    inp = synth_inputs(cfg, B, L, seed=seed + 50, lengths=lengths)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    return sd, inp, ref


CASES = {
    "c2arch_ragged": (lambda: preset("c2"), 4, 64, [64, 50, 33, 7], dict(duration_bias=1.5)),
    "refdefault_dw": (lambda: preset("ref-default"), 3, 48, [48, 30, 11], dict(duration_bias=1.4)),
    "ls_h768_2layer": (lambda: Fs2Config(**{**preset("c3").to_dict(), "encoder_layers": 1, "decoder_layers": 2,
                                            "variance_nlayers": [2, 2, 2]}), 2, 40, [40, 22], dict(duration_bias=1.3)),
    "h1024_dense_1layer": (lambda: Fs2Config(**{**preset("c5").to_dict(), "encoder_layers": 1, "decoder_layers": 1,
                                                "variance_nlayers": [1, 1, 1], "duration_nlayers": 1}),
                           2, 24, [24, 9], dict(duration_bias=1.3)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_fp32_matches_oracle(case):
    mk, B, L, lengths, skw = CASES[case]
    cfg = mk()
    sd, inp, ref = _oracle_case(cfg, B, L, lengths, seed=3, **skw)
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    out = _cpu(m(batch, inference=True))
    flips = int((out["duration_rounded"] != ref["duration_rounded"]).sum())
    errs = {"duration_prediction": float((out["duration_prediction"] - ref["duration_prediction"]).abs().max())}
    errs["encoder_out"] = float((m.engine.debug_tensor("encoder_out").cpu() - ref["_intermediates"]["encoder_out"]).abs().max())
    _report(test="oracle_fp32", case=case, duration_flips=flips, errs=errs)
    assert errs["encoder_out"] <= MEL_TOL_FP32
    if flips:  # the oracle itself sat within float noise of .5: compare under its durations instead
        out = _cpu(m.forward({"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])},
                             force_durations=ref["duration_rounded"]))
    assert torch.equal(out["tgt_mask"], ref["tgt_mask"]) and torch.equal(out["src_mask"], ref["src_mask"])
    bflips = {}
    for v in cfg.variances:
        bflips[v] = int((m.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref["_intermediates"][f"bucket_{v}"]).sum())
        errs[f"variances_{v}"] = float((out[f"variances_{v}"] - ref[f"variances_{v}"]).abs().max())
    errs["mel"] = float((out["mel"] - ref["mel"]).abs().max())
    _report(test="oracle_fp32", case=case, bucket_flips=bflips, errs=errs)
    if sum(bflips.values()):  # the oracle sat within float noise of a bin edge (reported above):
        out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],   # compare under ITS decisions
                             force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
        errs["mel"] = float((out["mel"] - ref["mel"]).abs().max())
    assert errs["mel"] <= MEL_TOL_FP32, errs


@pytest.mark.parametrize("case", list(CASES))
def test_fp32x3_matches_oracle_at_the_fp32_tolerance(case):
    """precision="fp32x3" (FS2_F32_X3): the fp32 mode with EVERY GEMM / conv as bf16 x 3 split products of the fp32 operands.  Held
    to the fp32 mode's own bar against the oracle (MEL_TOL_FP32 = 1e-3 on the encoder output and on the mel under the oracle's
    decisions); duration / bucket flips against the oracle are reported like the fp32 mode's."""
    mk, B, L, lengths, skw = CASES[case]
    cfg = mk()
    sd, inp, ref = _oracle_case(cfg, B, L, lengths, seed=3, **skw)
    m = _model(cfg, sd, "fp32x3")
    m.engine.set_debug(True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    out = _cpu(m(batch, inference=True))
    dflips = int((out["duration_rounded"] != ref["duration_rounded"]).sum())
    enc = float((m.engine.debug_tensor("encoder_out").cpu() - ref["_intermediates"]["encoder_out"]).abs().max())
    assert enc <= MEL_TOL_FP32, enc
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"]))
    assert torch.equal(out["tgt_mask"], ref["tgt_mask"]) and torch.equal(out["src_mask"], ref["src_mask"])
    bflips = {v: int((m.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref["_intermediates"][f"bucket_{v}"]).sum())
              for v in cfg.variances}
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                         force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    errs = {f"variances_{v}": float((out[f"variances_{v}"] - ref[f"variances_{v}"]).abs().max()) for v in cfg.variances}
    errs["mel"] = float((out["mel"] - ref["mel"]).abs().max())
    _report(test="oracle_fp32x3", case=case, duration_flips=dflips, bucket_flips=bflips, encoder_out_max=enc, errs=errs)
    assert errs["mel"] <= MEL_TOL_FP32, errs
    assert dflips <= max(1, ref["duration_rounded"].numel() // 100) and sum(bflips.values()) <= max(2, sum(ref["_intermediates"][f"bucket_{v}"].numel() for v in cfg.variances) // 50)


@pytest.mark.parametrize("mode", ["fp32x3", "mixed3"])
def test_split_arithmetic_predictor_single_launch_against_its_layer_launches(mode):
    """r05: in the split-arithmetic engines a dense 256-channel predictor is ONE launch (predictor_fused_kernel<..., X3>: activations as
    bf16 heads + tails in LDS, three MFMAs per product, fp32 LayerNorm) instead of a conv + LayerNorm launch per layer
    (fs2_set_fused_predictor(0)).  Same split products, another order of the K sum: the predictions agree to fp32 rounding of that
    order, the decisions with the oracle's as often, and both sit at the fp32 bar against the oracle."""
    mk, B, L, lengths, skw = CASES["c2arch_ragged"]
    cfg = mk()
    sd, inp, ref = _oracle_case(cfg, B, L, lengths, seed=3, **skw)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    forced = dict(force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    m = _model(cfg, sd, mode)
    a_free = _cpu(m(batch, inference=True))
    a = _cpu(m.forward(batch, **forced))
    m.engine.set_fused_predictor(False)
    b_free = _cpu(m(batch, inference=True))
    b = _cpu(m.forward(batch, **forced))
    m.engine.set_fused_predictor(True)
    rep = {"duration_prediction": float((a["duration_prediction"] - b["duration_prediction"]).abs().max())}
    assert rep["duration_prediction"] <= 2e-4
    for v in cfg.variances:
        k = f"variances_{v}"
        rep[k] = [float((a[k] - b[k]).abs().max()), float((a[k] - ref[k]).abs().max()), float((b[k] - ref[k]).abs().max())]
        assert rep[k][0] <= 2e-4 * (float(ref[k].abs().max()) + 1), (k, rep[k])
        assert rep[k][1] <= 1e-3, (k, rep[k])
    dfa = int((a_free["duration_rounded"] != ref["duration_rounded"]).sum())
    dfb = int((b_free["duration_rounded"] != ref["duration_rounded"]).sum())
    rep["duration_flips_single_vs_layers"] = [dfa, dfb]
    _report(test="x3_predictor_single_launch", mode=mode, **rep)
    assert dfa <= dfb + 1
    if mode == "fp32x3":
        assert float((a["mel"] - ref["mel"]).abs().max()) <= MEL_TOL_FP32


@pytest.mark.parametrize("case", ["c2arch_ragged", "refdefault_dw"])
def test_bf16_close_under_forced_durations(case):
    mk, B, L, lengths, skw = CASES[case]
    cfg = mk()
    sd, inp, ref = _oracle_case(cfg, B, L, lengths, seed=3, **skw)
    m = _model(cfg, sd, "bf16")
    m.engine.set_debug(True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    free = _cpu(m(batch, inference=True))
    dflips = int((free["duration_rounded"] != ref["duration_rounded"]).sum())
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"]))
    assert torch.equal(out["tgt_mask"], ref["tgt_mask"])
    bflips = {v: int((m.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref["_intermediates"][f"bucket_{v}"]).sum())
              for v in cfg.variances}
    verr = {v: float((out[f"variances_{v}"] - ref[f"variances_{v}"]).abs().max()) for v in cfg.variances}
    # bf16 noise on the predictions (~1e-2) is of the order of a bin (6/254): buckets flip, and with
    # random-init embeddings a flipped bucket is an unrelated row.  Judge the arithmetic with the
    # oracle's decisions forced (SURVEY §0.9 / §7 hard parts) and report the free-running flips.
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                         force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    err = (out["mel"] - ref["mel"]).abs()
    enc = float((m.engine.debug_tensor("encoder_out").cpu() - ref["_intermediates"]["encoder_out"]).abs().max())
    _report(test="bf16_forced", case=case, duration_flips_free=dflips, bucket_flips=bflips, variance_pred_max=verr,
            mel_max=float(err.max()), mel_mean=float(err.mean()), encoder_out_max=enc,
            mel_scale=float(ref["mel"].abs().max()))
    assert torch.isfinite(out["mel"]).all()
    assert enc <= BF16_ENC_MAX              # bf16 rounding (2^-8) through 4 post-LN layers of O(1) activations
    assert float(err.mean()) <= BF16_MEL_MEAN and float(err.max()) <= BF16_MEL_MAX  # bf16 tolerance, NOT the 1e-3 claim


def _decisions(m, cfg, batch, ref):
    """(duration flips, bucket flips under the oracle's durations, mel max-abs under all of the oracle's decisions)"""
    m.engine.set_debug(True)
    free = _cpu(m(batch, inference=True))
    dflips = int((free["duration_rounded"] != ref["duration_rounded"]).sum())
    _cpu(m.forward(batch, force_durations=ref["duration_rounded"]))
    bflips = sum(int((m.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref["_intermediates"][f"bucket_{v}"]).sum())
                 for v in cfg.variances)
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                         force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    return dflips, bflips, float((out["mel"] - ref["mel"]).abs().max()), free


@pytest.mark.parametrize("mode", ["mixed", "mixed3"])
@pytest.mark.parametrize("case", ["c2arch_ragged", "refdefault_dw", "ls_h768_2layer"])
def test_mixed_precision_is_decision_safe(case, mode):
    """precision="mixed" (FS2_MIXED): the encoder -> duration path and the variance-predictor chain in fp32, the decoder in
    bf16.  Its discrete decisions must be the fp32 path's: durations equal to the oracle's, bucket flips >= 10x rarer than
    the all-bf16 mode's (SURVEY 7 "hard parts": keep every predictor head and what feeds it out of bf16)."""
    mk, B, L, lengths, skw = CASES[case]
    cfg = mk()
    sd, inp, ref = _oracle_case(cfg, B, L, lengths, seed=3, **skw)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    d16, b16, e16, _ = _decisions(_model(cfg, sd, "bf16"), cfg, batch, ref)
    dmx, bmx, emx, free = _decisions(_model(cfg, sd, mode), cfg, batch, ref)
    nb = sum(ref["_intermediates"][f"bucket_{v}"].numel() for v in cfg.variances)
    _report(test="mixed_vs_bf16", case=case, mode=mode, buckets=nb, bf16=dict(duration_flips=d16, bucket_flips=b16, mel_forced=e16),
            mixed=dict(duration_flips=dmx, bucket_flips=bmx, mel_forced=emx))
    assert dmx == 0 and torch.equal(free["tgt_mask"], ref["tgt_mask"])
    assert bmx <= max(2, b16 // 10)
    assert emx <= BF16_MEL_MAX  # the decoder is bf16: same tolerance as the all-bf16 mode under forced decisions
    for v in cfg.variances[:1]:  # the first predictor sees fp32 inputs identical to the parity mode's
        assert float((free[f"variances_{v}"] - ref[f"variances_{v}"]).abs().max()) <= 1e-3


@pytest.mark.parametrize("mode", ["mixed", "mixed3"])
@pytest.mark.parametrize("name", [n for n in golden_names() if "teacher" not in n])
def test_mixed_precision_decisions_on_goldens(name, mode):
    g = Golden(name)
    m = _model(g.cfg, g.state_dict(), mode)
    batch = {"phones": torch.from_numpy(g.phones), "speaker": torch.from_numpy(g.speaker), **g.priors}
    out = _cpu(m(batch, inference=True))
    for k in ("duration_rounded", "src_mask", "tgt_mask"):
        assert np.array_equal(out[k].numpy(), g.out[k]), k
    err = float(np.abs(out["mel"].numpy() - g.out["mel"]).max())
    _report(test="golden_mixed", case=name, mode=mode, mel_max=err, mel_scale=float(np.abs(g.out["mel"]).max()))
    assert np.isfinite(err)


@pytest.mark.parametrize("cwt", [False, True], ids=["plain", "cwt"])
def test_deferred_layernorm_matches_its_own_launches(cwt):
    """hidden 768, depth-wise blocks: LayerNorm deferred into its consumers (pre-norm rows + row statistics from the GEMM
    epilogue; depth-wise conv / residual add normalise on load; normalise-only passes) against one launch per LayerNorm,
    and against the oracle, in fp32; same decisions."""
    mk, B, L, lengths, skw = CASES["ls_h768_2layer"]
    cfg = mk()
    if cwt:
        d = cfg.to_dict()
        d["variance_transforms"] = ["cwt", "none", "none"]
        d["stats"] = {**d["stats"], "pitch": {"min": 0.2, "max": 5.0, "mean": 0.0, "std": 1.0}}
        cfg = Fs2Config(**d)
    sd, inp, ref = _oracle_case(cfg, B, L, lengths, seed=3, **skw)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    m = _model(cfg, sd, "fp32")
    a = _cpu(m(batch, inference=True))
    m.engine.set_deferred_layernorm(False)
    b = _cpu(m(batch, inference=True))
    assert torch.equal(a["duration_rounded"], b["duration_rounded"]) and torch.equal(a["duration_rounded"], ref["duration_rounded"])
    d_ab = float((a["mel"] - b["mel"]).abs().max())
    d_ref = float((a["mel"] - ref["mel"]).abs().max())
    _report(test="deferred_ln", cwt=cwt, mel_deferred_vs_launches=d_ab, mel_deferred_vs_oracle=d_ref)
    assert d_ab <= 2e-4 and d_ref <= MEL_TOL_FP32
    if cwt:
        for k in ("spectrogram", "mean", "std", "reconstructed_signal"):
            assert float((a["variances_pitch"][k] - ref["variances_pitch"][k]).abs().max()) <= 1e-3, k
    m16 = _model(cfg, sd, "bf16")
    x = m16.forward(batch, force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    m16.engine.set_deferred_layernorm(False)
    y = m16.forward(batch, force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    assert float((x["mel"] - y["mel"]).abs().max()) <= BF16_MEL_MAX and float((x["mel"].cpu() - ref["mel"]).abs().mean()) <= BF16_MEL_MEAN


def test_full_size_properties_bf16():
    """BASELINE.json configs[1]: FS2-27M, batch 32 x 256 phonemes, 6 frames/phone -> T = 1536."""
    cfg = preset("c2")
    sd = synth_state_dict(cfg, 0, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0)
    inp = synth_inputs(cfg, 32, 256, seed=1234)
    m = _model(cfg, sd, "bf16")
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    out = m(batch, inference=True)
    assert tuple(out["mel"].shape) == (32, 1536, 80)
    assert bool((out["duration_rounded"] == 6).all()) and not bool(out["tgt_mask"].any())
    assert bool(torch.isfinite(out["mel"]).all())
    # idempotence: the same input gives the same bits
    again = m(batch, inference=True)
    assert torch.equal(out["mel"], again["mel"])
    # shard == whole: utterances are independent and (without pads) every shard sees the same T,
    # so a data-parallel shard reproduces its rows of the whole batch bit for bit
    half = m({"phones": batch["phones"][8:16], "speaker": batch["speaker"][8:16]}, inference=True)
    assert torch.equal(half["mel"], out["mel"][8:16])
    # speaker linearity probe: a different d-vector changes only that utterance
    spk2 = batch["speaker"].clone()
    spk2[3] = -spk2[3]
    out2 = m({"phones": batch["phones"], "speaker": spk2}, inference=True)
    same = [bool(torch.equal(out2["mel"][b], out["mel"][b])) for b in range(32)]
    assert same.count(False) == 1 and not same[3]


@pytest.mark.parametrize("case", ["c2arch_ragged", "c2_full", "c2_priorless_tail_rows", "c2_phone_level", "refdefault_phone_level"])
def test_variance_encoder_in_the_predictor_launch_is_bit_identical(case):
    """bf16 engine: bucketize + embedding add (+ pe + spk after the last variance) as the tail of the single-launch predictor
    (predictor_fused.hip, knob 1321 = default) against the stand-alone bucket_embed launch (knob 1320): same arithmetic, same
    bits - mel, every variance prediction, masks; ragged utterances (pad rows bucketize the masked 0), tile seams, T not a
    multiple of the tile's finished rows."""
    from lightningfastspeech2_amd import _lib
    cfg = preset("c2")
    if case.endswith("phone_level"):  # r06: phone-level variances ride in the ENCODE phase's predictor launches (dense / depth-wise)
        base = preset("c2" if case.startswith("c2") else "ref-default").to_dict()
        cfg = Fs2Config(**{**base, "variance_levels": ["phone", "frame", "phone"]})
        sd = synth_state_dict(cfg, 6, randomize_norm=True, duration_bias=1.45)
        inp = synth_inputs(cfg, 3, 150, seed=91, lengths=[150, 77, 149])
    elif case == "c2_full":
        sd = synth_state_dict(cfg, 0, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0, randomize_norm=True)
        inp = synth_inputs(cfg, 8, 256, seed=1234)
    elif case == "c2arch_ragged":
        sd = synth_state_dict(cfg, 3, randomize_norm=True, duration_bias=1.5)
        inp = synth_inputs(cfg, 4, 64, seed=53, lengths=[64, 50, 33, 7])
    else:
        sd = synth_state_dict(cfg, 5, randomize_norm=True, duration_bias=1.2)
        inp = synth_inputs(cfg, 3, 97, seed=77, lengths=[97, 61, 1])
    m = _model(cfg, sd, "bf16")
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    outs = {}
    try:
        for knob in (1320, 1321, 1320, 1321):
            m.engine.set_tuning(knob)
            outs.setdefault(knob, []).append(_cpu(m(batch, inference=True)))
    finally:
        m.engine.set_tuning(1321)
    a, b = outs[1320][0], outs[1321][0]
    assert torch.isfinite(b["mel"]).all() and b["mel"].shape[1] > 0
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
            assert torch.equal(outs[1321][1][k], b[k]), k


def test_full_size_fp32_vs_oracle_one_utterance():
    """One full-length utterance (L=256 -> T=1536) of the FS2-27M config against the oracle."""
    cfg = preset("c2")
    sd = synth_state_dict(cfg, 0, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0)
    inp = synth_inputs(cfg, 1, 256, seed=1234)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    out = _cpu(m({"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}, inference=True))
    assert torch.equal(out["duration_rounded"], ref["duration_rounded"])
    bflips = {v: int((m.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref["_intermediates"][f"bucket_{v}"]).sum())
              for v in cfg.variances}
    err = float((out["mel"] - ref["mel"]).abs().max())
    # at T=1536 x 3 variances x 255 edges a ~5e-6 prediction difference lands on the other side of
    # an edge for a frame or two, and the swapped embedding row then feeds the next predictor:
    # count those, then hold mel to 1e-3 under the oracle's own decisions
    forced = _cpu(m.forward({"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])},
                            force_durations=ref["duration_rounded"],
                            force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    ferr = float((forced["mel"] - ref["mel"]).abs().max())
    _report(test="fullsize_fp32_1utt", mel_max_free=err, mel_max_forced=ferr, bucket_flips_free=bflips)
    assert ferr <= MEL_TOL_FP32
    if sum(bflips.values()) == 0:
        assert err <= MEL_TOL_FP32
    assert bflips[cfg.variances[0]] <= 3  # first predictor sees identical inputs: only near-tie flips


def test_full_size_full_batch_fp32_and_mixed_vs_oracle():
    """BASELINE configs[1] at its FULL size - all 32 utterances x 256 phonemes -> T = 1536 - against the oracle:
    fp32 parity mode holds mel <= 1e-3 on every entry of the (32, 1536, 80) tensor under the oracle's decisions (free-running
    it differs from the oracle in a handful of near-tie buckets, counted); the mixed modes reproduce the durations exactly,
    stay within a small multiple of the fp32 mode's bucket flips and hold the bf16-decoder tolerance."""
    cfg = preset("c2")
    sd = synth_state_dict(cfg, 0, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0)
    inp = synth_inputs(cfg, 32, 256, seed=1234)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    nb = sum(ref["_intermediates"][f"bucket_{v}"].numel() for v in cfg.variances)
    rep = {"buckets": nb}
    for mode in ("fp32", "fp32x3", "mixed", "mixed3", "bf16"):
        dfl, bfl, forced, free = _decisions(_model(cfg, sd, mode), cfg, batch, ref)
        rep[mode] = dict(duration_flips=dfl, bucket_flips=bfl, mel_forced=forced)
        assert dfl == 0 and tuple(free["mel"].shape) == (32, 1536, 80)
        if mode == "fp32":
            assert forced <= MEL_TOL_FP32 and bfl <= nb // 200   # ~0.3 % measured: near-tie buckets only
        elif mode == "fp32x3":  # the parity-grade mode at a third of the time: the fp32 bar on the whole tensor
            assert forced <= MEL_TOL_FP32 and bfl <= max(10 * rep["fp32"]["bucket_flips"], nb // 100)
        elif mode != "bf16":
            assert forced <= BF16_MEL_MAX and bfl <= max(10 * rep["fp32"]["bucket_flips"], nb // 50)
        else:
            assert forced <= BF16_MEL_MAX
    _report(test="fullsize_fullbatch", **rep)
    assert rep["mixed3"]["bucket_flips"] * 10 <= rep["bf16"]["bucket_flips"]


def _random_gpu_cfg(rs):
    """Random hparams inside the engine's envelope (hidden a multiple of 64, head dim 32 / 64 / 128): dense / depth-wise
    mixes, odd kernel sizes up to 25, 1-3 variances incl. the CWT head, predictor depths 1-4, with and without priors."""
    H = int(rs.choice([64, 128, 192, 256, 320]))
    heads = [h for h in (1, 2, 3, 4, 5, 6, 8, 10) if H % h == 0 and H // h in (32, 64, 128)]
    he, hd = int(rs.choice(heads)), int(rs.choice(heads))
    dw = [bool(rs.randint(2)) for _ in range(4)]
    nl_e, nl_d = int(rs.randint(1, 3)), int(rs.randint(1, 4))
    odd = lambda hi=25: int(rs.choice([k for k in (1, 3, 5, 7, 9, 13, 17, 21, 25) if k <= hi]))
    variances = list(rs.permutation(["pitch", "energy", "snr", "srmr"])[: rs.randint(1, 5)])  # up to FS2_MAX_VARIANCES, as the shipped recipe has
    nv = len(variances)
    cwt = [bool(rs.randint(3) == 0) for _ in range(nv)]
    stats = {}
    for v, c in zip(variances, cwt):
        stats[v] = ({"min": 0.3, "max": 4.0, "mean": 0.0, "std": 1.0} if c else
                    {"min": float(-1 - rs.rand()), "max": float(1 + 2 * rs.rand()), "mean": float(rs.randn() * .3), "std": float(.5 + rs.rand())})
    priors = list(rs.permutation(["pitch", "energy", "duration", "snr", "srmr"])[: rs.randint(1, 6)]) if rs.randint(3) == 0 else []
    for pr in priors:  # (scripts/train.sh:49 lists five)
        stats[f"{pr}_prior"] = {"min": -1.0, "max": 1.5}
    return Fs2Config(
        n_phones=int(rs.randint(5, 60)), encoder_hidden=H, decoder_hidden=H, encoder_head=he, decoder_head=hd,
        encoder_layers=nl_e, decoder_layers=nl_d, encoder_kernel_sizes=[odd() for _ in range(nl_e)],
        decoder_kernel_sizes=[odd() for _ in range(nl_d)], encoder_depthwise_conv=dw[0], decoder_depthwise_conv=dw[1],
        encoder_conv_filter_size=H * int(rs.choice([1, 2, 4])), decoder_conv_filter_size=H * int(rs.choice([1, 2, 4])),
        variances=variances, variance_levels=[str(rs.choice(["frame", "frame", "phone"])) for _ in range(nv)],
        variance_transforms=["cwt" if c else "none" for c in cwt],
        variance_nlayers=[int(rs.randint(1, 5)) for _ in range(nv)], variance_kernel_size=[odd(9) for _ in range(nv)],
        variance_filter_size=H, variance_nbins=int(rs.choice([8, 33, 256])), variance_depthwise_conv=dw[2],
        duration_nlayers=int(rs.randint(1, 3)), duration_kernel_size=odd(9), duration_filter_size=H,
        duration_depthwise_conv=dw[3], n_mels=int(rs.choice([8, 80])), priors=priors, stats=stats)


@pytest.mark.parametrize("seed", range(10))
def test_fp32_random_configs_vs_oracle(seed):
    """Fuzz: random architectures x ragged batches, fp32 engine vs the oracle under the oracle's decisions (<= 1e-3 on mel and
    on every intermediate), free-running decision flips counted; the same batch through the bf16 engine stays finite."""
    rs = np.random.RandomState(500 + seed)
    cfg = _random_gpu_cfg(rs)
    B, L = int(rs.randint(1, 6)), int(rs.randint(3, 70))
    lengths = [L] + [int(rs.randint(1, L + 1)) for _ in range(B - 1)]
    sd = synth_state_dict(cfg, seed, randomize_norm=True, duration_bias=float(rs.uniform(0.4, 1.6)))
    inp = synth_inputs(cfg, B, L, seed=seed, lengths=lengths)
    pri = {k: v for k, v in inp.items() if k.startswith("priors_")}
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True, priors=pri)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"]), **pri}
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    free = _cpu(m(batch, inference=True))
    dfl = int((free["duration_rounded"] != ref["duration_rounded"]).sum())
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                         force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    assert torch.equal(out["tgt_mask"], ref["tgt_mask"]) and torch.equal(out["src_mask"], ref["src_mask"])
    errs = {"mel": float((out["mel"] - ref["mel"]).abs().max()),
            "duration_prediction": float((out["duration_prediction"] - ref["duration_prediction"]).abs().max())}
    for k in ("encoder_out", "regulated", "adaptor_out", "decoder_out"):
        errs[k] = float((m.engine.debug_tensor(k).cpu() - ref["_intermediates"][k]).abs().max())
    _report(test="fuzz_fp32", seed=seed, H=cfg.hidden, dw=[cfg.encoder_depthwise_conv, cfg.decoder_depthwise_conv,
            cfg.variance_depthwise_conv, cfg.duration_depthwise_conv], transforms=cfg.variance_transforms[:len(cfg.variances)],
            B=B, L=L, T=int(ref["mel"].shape[1]), duration_flips_free=dfl, errs=errs)
    assert dfl <= 1 and all(e <= MEL_TOL_FP32 for e in errs.values()), errs
    if ref["mel"].shape[1] > 0:
        o16 = _model(cfg, sd, "bf16").forward(batch, force_durations=ref["duration_rounded"])
        assert bool(torch.isfinite(o16["mel"]).all())
        # the split arithmetic (every GEMM / conv / attention product as bf16 x 3) on the same architecture: the fp32 bar
        ox = _cpu(_model(cfg, sd, "fp32x3").forward(batch, force_durations=ref["duration_rounded"],
                                                     force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
        ex = float((ox["mel"] - ref["mel"]).abs().max())
        _report(test="fuzz_fp32x3", seed=seed, mel=ex)
        assert ex <= MEL_TOL_FP32, ex


def test_rejects_training_forward_and_bad_ids():
    g = Golden("dense_small")
    m = _model(g.cfg, g.state_dict(), "fp32")
    batch = {"phones": torch.from_numpy(g.phones), "speaker": torch.from_numpy(g.speaker)}
    with pytest.raises(KeyError):
        m(batch)  # inference=False is the teacher-forced forward: it needs targets["duration"], ["variances_*"]
    bad = torch.from_numpy(g.phones).clone()
    bad[0, 0] = g.cfg.n_phones
    with pytest.raises(IndexError):
        m({"phones": bad, "speaker": batch["speaker"]}, inference=True)


def test_one_engine_many_shapes_and_checkpoint_roundtrip(tmp_path):
    """The same engine serves batches of changing (B, L, T) (arena regrowth, output pre-allocation
    guess misses) and gives the same answer as a fresh engine; a Lightning-style checkpoint dict
    (state_dict + hyper_parameters + stats + phone2id, fastspeech2.py:622-634) loads through
    FastSpeech2.from_checkpoint."""
    from lightningfastspeech2_amd.model import FastSpeech2
    g = Golden("mid_dense_d128")
    sd = g.state_dict()
    ckpt = {
        "state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()},
        "hyper_parameters": {**{k: v for k, v in g.cfg.to_dict().items() if k not in ("stats", "n_phones")},
                             "lr": 1e-4, "fastdiff_variances": False},  # extra keys are ignored
        "stats": g.cfg.stats,
        "phone2id": {f"p{i}": i for i in range(g.cfg.n_phones)},
        "speaker2dvector": {"spk": g.speaker[0]},
    }
    path = tmp_path / "lit_model.ckpt"
    torch.save(ckpt, path)
    m = FastSpeech2.from_checkpoint(str(path), precision="fp32", device="cuda:0")
    assert m.hparams.encoder_hidden == g.cfg.hidden and len(m.phone2id) == g.cfg.n_phones
    out = _cpu(m({"phones": torch.from_numpy(g.phones), "speaker": torch.from_numpy(g.speaker)}, inference=True))
    assert float(np.abs(out["mel"].numpy() - g.out["mel"]).max()) <= MEL_TOL_FP32
    # other shapes through the same engine, then the first batch again
    for B, L, seed in ((5, 40, 1), (1, 7, 2), (3, 90, 3)):
        inp = synth_inputs(g.cfg, B, L, seed=seed, lengths=[L] + [max(1, L // (i + 2)) for i in range(B - 1)])
        ref = oracle_cpu.forward(sd, g.cfg, inp["phones"], inp["speaker"])
        o = _cpu(m({"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}, inference=True))
        assert tuple(o["mel"].shape) == tuple(ref["mel"].shape)
        # always compared: under the oracle's decisions (a near-tie duration / bucket may differ free-running) at 1e-3
        ref = oracle_cpu.forward(sd, g.cfg, inp["phones"], inp["speaker"], return_intermediates=True)
        b2 = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
        f = _cpu(m.forward(b2, force_durations=ref["duration_rounded"],
                           force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in g.cfg.variances}))
        assert torch.equal(f["tgt_mask"], ref["tgt_mask"])
        assert float((f["mel"] - ref["mel"]).abs().max()) <= MEL_TOL_FP32
        assert int((o["duration_rounded"] != ref["duration_rounded"]).sum()) <= 1
    again = _cpu(m({"phones": torch.from_numpy(g.phones), "speaker": torch.from_numpy(g.speaker)}, inference=True))
    assert torch.equal(again["mel"], out["mel"])


@pytest.mark.gpu
def test_decode_graph_replay_is_bit_identical():
    """fs2_set_graphs: both phases replayed as hipGraphs (first sight plain, second captured, then replays) give the plain
    path's outputs bit for bit - the encode phase's (durations, masks, the frame count the host reads between the phases) and
    the decode phase's - for two alternating output-buffer sets and after a shape change."""
    import numpy as np
    import torch
    from lightningfastspeech2_amd.config import preset
    from lightningfastspeech2_amd.model import FastSpeech2
    from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
    cfg = preset("ref-default")
    sd = synth_state_dict(cfg, 3, duration_bias=float(np.log(4.0)), duration_weight_scale=0.0)
    model = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
    model.engine.set_graphs(False)
    batches = {}

    def run(B, L, seed):
        if (B, L, seed) not in batches:  # the same input tensors every time: their addresses are part of the encode signature
            inp = synth_inputs(cfg, B, L, seed=seed)
            batches[(B, L, seed)] = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}
        out = model(batches[(B, L, seed)], inference=True)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in out.items() if isinstance(v, torch.Tensor)}

    want_a, want_b = run(3, 40, 1), run(2, 24, 2)
    model.engine.set_graphs(True)
    n0 = model.engine.graph_replays()
    keep = []
    for it in range(6):
        got = run(3, 40, 1)
        keep.append(got)  # hold the outputs: the allocator hands out different buffers, more than one signature is cached
        keep = keep[-2:]
        for k, v in want_a.items():
            assert torch.equal(got[k], v), (it, k)
    n1 = model.engine.graph_replays()
    assert n1 >= n0 + 5  # the encode phase of every forward after the first (its inputs stay put); the decode phase whenever the allocator repeats an output set
    for it in range(4):
        got = run(2, 24, 2)
        for k, v in want_b.items():
            assert torch.equal(got[k], v), (it, k)
    assert model.engine.graph_replays() >= n1 + 3 + 1  # encode from the second forward on, decode once its output set repeats
    got = run(3, 40, 1)
    for k, v in want_a.items():
        assert torch.equal(got[k], v), k
    model.engine.set_graphs(False)
    got = run(3, 40, 1)
    for k, v in want_a.items():
        assert torch.equal(got[k], v), k
