"""Full-size parity corners (`-m gpu`, VERDICT r04 "what's weak" 1-3):

(a) full-LENGTH RAGGED inference against the oracle - SURVEY 8d's ragged variant (phone lengths ~U{128..256}, 0-padded, a random
    duration head: every utterance its own frame count, a real `tgt_mask`, key padding through the decoder's 384-query attention
    items, pad rows feeding the conv halos of valid rows - fastspeech2.py:651, model.py:349-370) for BASELINE configs[1] (B = 8) and
    configs[2] (B = 4): fp32 and fp32x3 at the north star's 1e-3, bf16 / mixed3 under the oracle's decisions at a bf16 tolerance
    that is <= 3.5x what was measured;
(b) the NEAR-TIE claim every full-size 1e-3 assertion rests on: where the fp32 engine, free-running, puts a frame into another
    bucket than the oracle (or rounds a duration the other way), the ORACLE's value sits within float noise of that decision's
    edge - or the flip lies inside the receptive-field cone of an earlier variance's flip, whose swapped embedding row the later
    predictor legitimately sees (model.py:315-333: each frame-level encoder adds its embedding before the next predictor runs).
    Nothing else may differ.
"""
import functools

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
from oracle import oracle_cpu
from test_gpu_forward import MEL_TOL_FP32, _cpu, _model, _report

pytestmark = pytest.mark.gpu

# bf16 under the oracle's decisions, measured r04 on ragged / full-size batches: mel max-abs 0.014-0.018, mean 0.0023-0.0030 on
# O(1)-scale mels (scale 2.5-2.8).  The asserted bounds are <= 3.5x that (they were 15-20x until r04).
BF16_MEL_MAX, BF16_MEL_MEAN = 0.06, 0.008

RAGGED = {"c2": 8, "c3": 4}  # utterances in the ragged full-length batch


@functools.lru_cache(maxsize=2)
def _ragged_case(name):
    """lengths ~ U{128..256} (one utterance at the full 256), duration head = half-scale random weights around ln 6:
    durations 3..8 per phone, T of the longest utterance ~1300-1450 frames, well inside the 2756-frame clip."""
    cfg = preset(name)
    B = RAGGED[name]
    rs = np.random.RandomState(20 + B)
    lengths = [256] + [int(rs.randint(128, 257)) for _ in range(B - 1)]
    sd = synth_state_dict(cfg, 4, randomize_norm=True, duration_bias=float(np.log(6.0)), duration_weight_scale=0.5)
    inp = synth_inputs(cfg, B, 256, seed=4321, lengths=lengths)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    return cfg, sd, inp, ref, lengths


def _bin_edges(sd, var):
    return np.asarray(sd[f"variance_adaptor.encoders.{var}.bins"], np.float64)


def _edge_distance(values, edges):
    """distance of each value to the nearest bucket edge"""
    v = np.asarray(values, np.float64)
    j = np.clip(np.searchsorted(edges, v), 1, len(edges) - 1)
    return np.minimum(np.abs(v - edges[j - 1]), np.abs(v - edges[j]))


def _near_tie_report(cfg, sd, ref, got_buckets, got_durations=None):
    """Classify every decision the engine made differently from the oracle.  Returns a dict with, per variance, the number of
    flips, how many of them are ROOT near-ties (the oracle's value within float noise of an edge) and how many lie in the cone
    of an earlier variance's flip; `unexplained` must be 0."""
    rep = {"unexplained": 0, "max_root_edge_distance": 0.0}
    B, T = ref["tgt_mask"].shape
    earlier = np.zeros((B, T), bool)  # frames whose row of x already differs from the oracle's (a swapped embedding row upstream)
    for vi, var in enumerate(cfg.variances):
        st = cfg.stats[var]
        ref_idx = ref["_intermediates"][f"bucket_{var}"].numpy()
        flips = got_buckets[var] != ref_idx
        val = ref[f"variances_{var}"].numpy().astype(np.float64) * float(st["std"]) + float(st["mean"])  # model.py:434
        dist = _edge_distance(val, _bin_edges(sd, var))
        tol = 5e-5 * np.abs(val) + 1e-5
        near = dist <= tol
        # the predictor's receptive field: nlayers x (k - 1) / 2 frames each side (model.py:482-561), inside one utterance
        reach = int(cfg.variance_nlayers[vi]) * (int(cfg.variance_kernel_size[vi]) - 1) // 2
        cone = np.zeros_like(earlier)
        if earlier.any():
            for b, t in zip(*np.nonzero(earlier)):
                cone[b, max(0, t - reach):t + reach + 1] = True
        root = flips & near & ~cone
        down = flips & cone
        bad = flips & ~near & ~cone
        rep[var] = {"flips": int(flips.sum()), "root_near_ties": int(root.sum()), "in_cone_of_earlier_flip": int(down.sum()),
                    "unexplained": int(bad.sum())}
        if root.any():
            rep["max_root_edge_distance"] = max(rep["max_root_edge_distance"], float(dist[root].max()))
        rep["unexplained"] += int(bad.sum())
        earlier = earlier | flips  # x changes only where an embedding row was swapped (x = x + emb[idx], model.py:333); the cone is where a predictor SEES it
    if got_durations is not None:
        dp = ref["duration_prediction"].numpy().astype(np.float64)
        x = np.exp(dp) - 1.0                                   # model.py:300: round(exp(p) - 1)
        dflip = got_durations != ref["duration_rounded"].numpy()
        half = np.abs(x - (np.floor(x) + 0.5))
        ok = half <= 5e-5 * np.abs(x) + 1e-5
        rep["duration"] = {"flips": int(dflip.sum()), "unexplained": int((dflip & ~ok).sum())}
        rep["unexplained"] += rep["duration"]["unexplained"]
    return rep


def _free_buckets(m, cfg):
    return {v: m.engine.debug_tensor(f"bucket_{v}").cpu().long().numpy() for v in cfg.variances}


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_full_length_ragged_fp32_vs_oracle(name):
    cfg, sd, inp, ref, lengths = _ragged_case(name)
    B = RAGGED[name]
    T = int(ref["mel"].shape[1])
    totals = (~ref["tgt_mask"]).sum(1)
    assert 900 <= T <= 2756 and int(totals.min()) < T - 100, (T, totals.tolist())   # ragged in frames too, no clip
    assert bool(ref["src_mask"].any()) and bool(ref["tgt_mask"].any())
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    rep = {"T": T, "lengths": lengths, "frames": totals.tolist()}
    for mode in ("fp32", "fp32x3"):
        m = _model(cfg, sd, mode)
        m.engine.set_debug(True)
        free = _cpu(m(batch, inference=True))
        enc = float((m.engine.debug_tensor("encoder_out").cpu() - ref["_intermediates"]["encoder_out"]).abs().max())
        dpe = float((free["duration_prediction"] - ref["duration_prediction"]).abs().max())
        got_d = free["duration_rounded"].numpy()
        # buckets under the oracle's durations (the same T): the engine's own decisions on the oracle's frames
        out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"]))
        assert torch.equal(out["tgt_mask"], ref["tgt_mask"]) and torch.equal(out["src_mask"], ref["src_mask"])
        got_b = _free_buckets(m, cfg)
        nt = _near_tie_report(cfg, sd, ref, got_b, got_d)
        forced = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                                force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
        errs = {"mel": float((forced["mel"] - ref["mel"]).abs().max()),
                "decoder_out": float((m.engine.debug_tensor("decoder_out").cpu() - ref["_intermediates"]["decoder_out"]).abs().max()),
                "adaptor_out": float((m.engine.debug_tensor("adaptor_out").cpu() - ref["_intermediates"]["adaptor_out"]).abs().max())}
        for v in cfg.variances:
            errs[f"variances_{v}"] = float((forced[f"variances_{v}"] - ref[f"variances_{v}"]).abs().max())
        rep[mode] = {"encoder_out": enc, "duration_prediction": dpe, "near_tie": nt, "errs_forced": errs}
        assert enc <= MEL_TOL_FP32 and dpe <= MEL_TOL_FP32, (mode, enc, dpe)
        assert all(e <= MEL_TOL_FP32 for e in errs.values()), (mode, errs)          # every entry of (B, T, 80), pad rows included
        if mode == "fp32":
            # whatever the parity mode decides differently from the oracle is a near-tie or its downstream cone - nothing else
            assert nt["unexplained"] == 0, nt
            nb = B * T
            assert nt[cfg.variances[0]]["flips"] <= max(3, nb // 300), nt            # the first predictor sees the oracle's inputs
            if all(nt[v]["flips"] == 0 for v in cfg.variances) and nt["duration"]["flips"] == 0:
                assert float((free["mel"] - ref["mel"]).abs().max()) <= MEL_TOL_FP32
        else:  # the split arithmetic is ~1e-5 from fp32: a few more near-ties, durations still the oracle's
            assert nt["duration"]["flips"] <= 1, nt
    _report(test="ragged_full_length", case=name, **rep)


@pytest.mark.parametrize("name", ["c2", "c3"])
@pytest.mark.parametrize("mode", ["bf16", "mixed3"])
def test_full_length_ragged_bf16_under_the_oracles_decisions(name, mode):
    cfg, sd, inp, ref, lengths = _ragged_case(name)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    m = _model(cfg, sd, mode)
    free = _cpu(m(batch, inference=True))
    dfl = int((free["duration_rounded"] != ref["duration_rounded"]).sum())
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                         force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    assert torch.equal(out["tgt_mask"], ref["tgt_mask"]) and torch.isfinite(out["mel"]).all()
    err = (out["mel"] - ref["mel"]).abs()
    valid = ~ref["tgt_mask"]
    _report(test="ragged_full_length_bf16", case=name, mode=mode, duration_flips_free=dfl, mel_max=float(err.max()), mel_mean=float(err.mean()),
            mel_max_valid_frames=float(err[valid].max()), mel_scale=float(ref["mel"].abs().max()))
    assert float(err.max()) <= BF16_MEL_MAX and float(err.mean()) <= BF16_MEL_MEAN, (float(err.max()), float(err.mean()))
    if mode == "mixed3":  # the decision-safe mode: the front is fp32-grade, so the durations are the oracle's
        assert dfl == 0


def test_near_tie_claim_at_baseline_size_full_batch():
    """BASELINE configs[1] at its full size (32 x 256 phonemes -> T = 1536, the timed workload): every bucket the fp32 engine
    assigns differently from the oracle (343 of 147 456 measured in r04) is a root near-tie or lies in the cone of one."""
    cfg = preset("c2")
    sd = synth_state_dict(cfg, 0, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0)
    inp = synth_inputs(cfg, 32, 256, seed=1234)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    out = _cpu(m(batch, inference=True))
    assert torch.equal(out["duration_rounded"], ref["duration_rounded"])
    nt = _near_tie_report(cfg, sd, ref, _free_buckets(m, cfg))
    _report(test="near_tie_fullsize", **nt)
    assert nt["unexplained"] == 0, nt
    assert nt[cfg.variances[0]]["flips"] == nt[cfg.variances[0]]["root_near_ties"]  # nothing upstream of the first predictor differs
    assert nt["max_root_edge_distance"] <= 1e-4
