"""BASELINE.json configs[2] (LS-76M, batch 32) and the per-GPU share of configs[4] (FS2-1B, batch 8) at their FULL
configuration (`-m gpu`): all layers, H = 768 / 1024, 256 phonemes -> T = 1536.

Per config: (a) the size-independent properties at the full per-GPU batch in the bf16 throughput mode - shape, finiteness,
bit-equal reruns, shard == whole (the data-parallel invariant), a speaker change touches exactly one utterance; (b) one
full-length utterance in fp32 parity mode against the CPU oracle, mel <= 1e-3 under the oracle's decisions, free-running
flips reported (a 5e-6 prediction difference next to one of 255 bucket edges flips a frame or two at this size, as it would
between two CPUs - see test_gpu_forward.test_full_size_fp32_vs_oracle_one_utterance); (c) r06: the FULL per-GPU batch against the
oracle - every entry of the (32, 1536, 80) / (8, 1536, 80) mel in fp32 at 1e-3 and in bf16 at the bf16 tolerance under the oracle's
decisions, the free-running flips classified (0 unexplained), plus a ragged full-length C5 batch.
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

FULL = {"c3": 32, "c5": 8}  # utterances per GPU (configs[3] = 256 / 8 GPUs, configs[4] = 64 / 8 GPUs)


@functools.lru_cache(maxsize=1)
def _weights(name):
    cfg = preset(name)
    return cfg, synth_state_dict(cfg, 0, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0)


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_full_config_properties_bf16(name):
    cfg, sd = _weights(name)
    B = FULL[name]
    inp = synth_inputs(cfg, B, 256, seed=1234)
    m = _model(cfg, sd, "bf16")
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    out = m(batch, inference=True)
    assert tuple(out["mel"].shape) == (B, 1536, 80)
    assert bool((out["duration_rounded"] == 6).all()) and not bool(out["tgt_mask"].any())
    assert bool(torch.isfinite(out["mel"]).all())
    for v in cfg.variances:
        assert bool(torch.isfinite(out[f"variances_{v}"]).all())
    again = m(batch, inference=True)
    assert torch.equal(out["mel"], again["mel"])
    lo, hi = B // 4, B // 2
    part = m({"phones": batch["phones"][lo:hi], "speaker": batch["speaker"][lo:hi]}, inference=True)
    assert torch.equal(part["mel"], out["mel"][lo:hi])
    spk2 = batch["speaker"].clone()
    spk2[1] = -spk2[1]
    out2 = m({"phones": batch["phones"], "speaker": spk2}, inference=True)
    same = [bool(torch.equal(out2["mel"][b], out["mel"][b])) for b in range(B)]
    assert same.count(False) == 1 and not same[1]
    # (the in-place wide-row LayerNorm epilogue - knob 301, off by default - has its own operator-level test with an asserted
    # tolerance, tests/test_gpu_ops.py::test_wide_layernorm_fused_vs_two_launches; a free-running whole-model comparison flips
    # buckets and asserted nothing)
    _report(test="full_config_bf16", case=name, mel_scale=float(out["mel"].abs().max()))


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_full_config_fp32_one_utterance_vs_oracle(name):
    cfg, sd = _weights(name)
    inp = synth_inputs(cfg, 1, 256, seed=1234)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    out = _cpu(m(batch, inference=True))
    assert torch.equal(out["duration_rounded"], ref["duration_rounded"])
    enc = float((m.engine.debug_tensor("encoder_out").cpu() - ref["_intermediates"]["encoder_out"]).abs().max())
    bflips = {v: int((m.engine.debug_tensor(f"bucket_{v}").cpu().long() != ref["_intermediates"][f"bucket_{v}"]).sum())
              for v in cfg.variances}
    err = float((out["mel"] - ref["mel"]).abs().max())
    forced = _cpu(m.forward(batch, force_durations=ref["duration_rounded"],
                            force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances}))
    ferr = float((forced["mel"] - ref["mel"]).abs().max())
    dec = float((m.engine.debug_tensor("decoder_out").cpu() - ref["_intermediates"]["decoder_out"]).abs().max())
    _report(test="full_config_fp32_1utt", case=name, mel_max_free=err, mel_max_forced=ferr, bucket_flips_free=bflips,
            encoder_out_max=enc, decoder_out_max_forced=dec)
    assert enc <= MEL_TOL_FP32 and ferr <= MEL_TOL_FP32
    if sum(bflips.values()) == 0:
        assert err <= MEL_TOL_FP32
    assert bflips[cfg.variances[0]] <= 3  # first predictor sees identical inputs: only near-tie flips


def test_persistent_gemm_whole_model_is_bit_identical():
    """The LightSpeech block (H = 768, depth-wise, deferred LayerNorm) at a size whose GEMMs take the persistent kernel
    (gemm_persist.hip: 12 x 1536 frames = 96 row tiles x 3 .. 12 column tiles > 256 CUs) against the same model with every
    GEMM one tile per workgroup (knob 220): the deferred epilogue's pre-norm rows + row statistics, the residual normalised on
    load, ReLU, the predictor chain - the same bits in mel, variances and durations."""
    from lightningfastspeech2_amd.config import Fs2Config
    cfg = Fs2Config(**{**preset("c3").to_dict(), "encoder_layers": 1, "decoder_layers": 2, "variance_nlayers": [2, 2, 2]})
    sd = synth_state_dict(cfg, 2, randomize_norm=True, duration_bias=float(np.log(7.0)), duration_weight_scale=0.0)
    inp = synth_inputs(cfg, 12, 256, seed=99)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    m = _model(cfg, sd, "bf16")
    outs = {}
    try:
        for knob in (220, 221, 221):
            m.engine.set_tuning(knob)
            outs.setdefault(knob, []).append(_cpu(m(batch, inference=True)))
    finally:
        m.engine.set_tuning(221)
    a, b, c = outs[220][0], outs[221][0], outs[221][1]
    assert tuple(b["mel"].shape) == (12, 1536, 80) and bool(torch.isfinite(b["mel"]).all())
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
            assert torch.equal(b[k], c[k]), k


def test_predictor_head_from_the_epilogue_sums_against_the_normalise_pass():
    """Wide depth-wise predictors (C3 / C4): the last layer's LayerNorm + Linear head is computed from row sums the last GEMM's epilogue
    leaves (GemmArgs::head_out; persistent kernel) instead of a normalise pass over stored activations (knob 230: the pass).  Same predictions to fp32 rounding of a different summation order - the pass
    reads bf16-rounded activations, the epilogue the fp32 ones, so the epilogue form is the closer one to the oracle."""
    from lightningfastspeech2_amd.config import Fs2Config
    cfg = Fs2Config(**{**preset("c3").to_dict(), "encoder_layers": 2, "decoder_layers": 2, "variance_nlayers": [2, 2, 2]})
    sd = synth_state_dict(cfg, 7, randomize_norm=True, duration_bias=1.4)
    inp = synth_inputs(cfg, 3, 40, seed=62, lengths=[40, 22, 9])
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    forced = dict(force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    m = _model(cfg, sd, "bf16")
    a = _cpu(m.forward(batch, **forced))
    m.engine.set_tuning(230)
    b = _cpu(m.forward(batch, **forced))
    m.engine.set_tuning(231)
    rep = {}
    for v in cfg.variances:
        k = f"variances_{v}"
        ra = torch.as_tensor(ref[k]).float()
        ea, eb = (a[k] - ra).abs(), (b[k] - ra).abs()
        rep[v] = [float((a[k] - b[k]).abs().max()), float(ea.max()), float(eb.max()), float(ea.mean()), float(eb.mean())]
        assert float((a[k] - b[k]).abs().max()) <= 0.05 * (float(ra.abs().max()) + 1), v   # two bf16 evaluations of the same predictor
        assert float(ea.mean()) <= 1.25 * float(eb.mean()) + 1e-4, v                        # no further from the oracle than the pass
    _report(test="head_sums", per_variance_sums_vs_pass_max__err_sums_max__err_pass_max__err_sums_mean__err_pass_mean=rep)
    assert torch.equal(a["tgt_mask"], b["tgt_mask"])


def test_folded_layernorm_against_its_own_passes():
    """Inside a stack of wide depth-wise bf16 blocks a block's closing LayerNorm is folded into the next block's in-projection
    (fs2_set_folded_layernorm, default) instead of a normalise-only pass per block: the two differ by bf16 roundings only (the
    activations are rounded before instead of after the normalisation, the folded weights once) - held to the bf16 tolerance
    against each other and against the oracle under its decisions (shard == whole at the full configuration:
    test_full_config_properties_bf16)."""
    from lightningfastspeech2_amd.config import Fs2Config
    from test_gpu_forward import BF16_MEL_MAX, BF16_MEL_MEAN
    cfg = Fs2Config(**{**preset("c3").to_dict(), "encoder_layers": 3, "decoder_layers": 3, "variance_nlayers": [2, 2, 2]})
    sd = synth_state_dict(cfg, 6, randomize_norm=True, duration_bias=1.4)
    inp = synth_inputs(cfg, 3, 40, seed=61, lengths=[40, 22, 9])
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    forced = dict(force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    m = _model(cfg, sd, "bf16")
    a = _cpu(m.forward(batch, **forced))
    m.engine.set_folded_layernorm(False)
    b = _cpu(m.forward(batch, **forced))
    m.engine.set_folded_layernorm(True)
    c = _cpu(m.forward(batch, **forced))
    assert torch.equal(a["mel"], c["mel"])
    d_ab = (a["mel"] - b["mel"]).abs()
    e_a, e_b = (a["mel"] - ref["mel"]).abs(), (b["mel"] - ref["mel"]).abs()
    _report(test="folded_ln", folded_vs_passes_max=float(d_ab.max()), folded_vs_oracle=[float(e_a.max()), float(e_a.mean())],
            passes_vs_oracle=[float(e_b.max()), float(e_b.mean())])
    assert float(d_ab.max()) <= BF16_MEL_MAX
    assert float(e_a.max()) <= BF16_MEL_MAX and float(e_a.mean()) <= BF16_MEL_MEAN
    assert float(e_a.mean()) <= 1.5 * float(e_b.mean()) + 1e-4   # no worse than the materialised form


# ---- r06 (VERDICT r05 "what's weak" 2): the oracle at the FULL per-GPU batch of configs[2] / configs[4] --------------------------------
# The shapes the persistent GEMM, the folded LayerNorms and the head-sum epilogues actually run at (49152 rows at C3): every entry of the
# (B, 1536, 80) mel against the CPU oracle, not properties only.

def _oracle_full(name, B, lengths=None, ragged_head=False):
    cfg = preset(name)
    if ragged_head:
        sd = synth_state_dict(cfg, 4, randomize_norm=True, duration_bias=float(np.log(6.0)), duration_weight_scale=0.5)
    else:
        sd = _weights(name)[1]
    inp = synth_inputs(cfg, B, 256, seed=1234 if lengths is None else 4321, lengths=lengths)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"], return_intermediates=True)
    return cfg, sd, inp, ref


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_full_config_full_batch_vs_oracle(name):
    """BASELINE configs[2] at B = 32 and the per-GPU share of configs[4] at B = 8, 256 phonemes -> T = 1536, every utterance against the
    oracle: fp32 mode - durations equal, mel <= 1e-3 over the WHOLE (B, 1536, 80) tensor under the oracle's decisions, every free-running
    bucket flip a near-tie or in the cone of one (0 unexplained); bf16 (the timed mode: persistent GEMM, folded LayerNorms, head sums from
    the epilogue at C3) under the oracle's decisions at the bf16 tolerance."""
    from test_gpu_parity_corners import BF16_MEL_MAX, BF16_MEL_MEAN, _free_buckets, _near_tie_report
    B = FULL[name]
    cfg, sd, inp, ref = _oracle_full(name, B)
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    forced_kw = dict(force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    free = _cpu(m(batch, inference=True))
    assert tuple(free["mel"].shape) == (B, 1536, 80)
    assert torch.equal(free["duration_rounded"], ref["duration_rounded"]) and torch.equal(free["tgt_mask"], ref["tgt_mask"])
    nt = _near_tie_report(cfg, sd, ref, _free_buckets(m, cfg))
    forced = _cpu(m.forward(batch, **forced_kw))
    err32 = float((forced["mel"] - ref["mel"]).abs().max())
    enc = float((m.engine.debug_tensor("encoder_out").cpu() - ref["_intermediates"]["encoder_out"]).abs().max())
    del m
    torch.cuda.empty_cache()
    m16 = _model(cfg, sd, "bf16")
    out16 = _cpu(m16.forward(batch, **forced_kw))
    e16 = (out16["mel"] - ref["mel"]).abs()
    nb = B * 1536 * len(cfg.variances)
    _report(test="full_config_full_batch", case=name, batch=B, buckets=nb, fp32_mel_forced=err32, fp32_encoder_out=enc, near_tie=nt,
            bf16_mel_forced_max=float(e16.max()), bf16_mel_forced_mean=float(e16.mean()), mel_scale=float(ref["mel"].abs().max()))
    assert err32 <= MEL_TOL_FP32 and enc <= MEL_TOL_FP32
    assert nt["unexplained"] == 0, nt
    assert sum(nt[v]["flips"] for v in cfg.variances) <= nb // 100, nt
    assert float(e16.max()) <= BF16_MEL_MAX and float(e16.mean()) <= BF16_MEL_MEAN, (float(e16.max()), float(e16.mean()))


def test_c5_ragged_full_length_batch_vs_oracle():
    """configs[4]'s per-GPU batch (B = 8) RAGGED at full length: phone lengths [256] + U{128..256}, a random duration head (every utterance
    its own frame count, a real tgt_mask, key padding in the 8-head decoder attention) - fp32 at 1e-3 under the oracle's decisions with 0
    unexplained flips, bf16 at the bf16 tolerance."""
    from test_gpu_parity_corners import BF16_MEL_MAX, BF16_MEL_MEAN, _free_buckets, _near_tie_report
    rs = np.random.RandomState(28)
    lengths = [256] + [int(rs.randint(128, 257)) for _ in range(7)]
    cfg, sd, inp, ref = _oracle_full("c5", 8, lengths=lengths, ragged_head=True)
    T = int(ref["mel"].shape[1])
    assert bool(ref["tgt_mask"].any()) and bool(ref["src_mask"].any()) and 900 <= T <= 2756
    batch = {"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])}
    m = _model(cfg, sd, "fp32")
    m.engine.set_debug(True)
    free = _cpu(m(batch, inference=True))
    got_d = free["duration_rounded"].numpy()
    out = _cpu(m.forward(batch, force_durations=ref["duration_rounded"]))
    assert torch.equal(out["tgt_mask"], ref["tgt_mask"])
    nt = _near_tie_report(cfg, sd, ref, _free_buckets(m, cfg), got_d)
    forced_kw = dict(force_durations=ref["duration_rounded"], force_buckets={v: ref["_intermediates"][f"bucket_{v}"] for v in cfg.variances})
    forced = _cpu(m.forward(batch, **forced_kw))
    err32 = float((forced["mel"] - ref["mel"]).abs().max())
    del m
    torch.cuda.empty_cache()
    out16 = _cpu(_model(cfg, sd, "bf16").forward(batch, **forced_kw))
    e16 = (out16["mel"] - ref["mel"]).abs()
    _report(test="c5_ragged_full_length", T=T, lengths=lengths, fp32_mel_forced=err32, near_tie=nt, bf16_mel_forced_max=float(e16.max()),
            bf16_mel_forced_mean=float(e16.mean()))
    assert err32 <= MEL_TOL_FP32 and nt["unexplained"] == 0, (err32, nt)
    assert float(e16.max()) <= BF16_MEL_MAX and float(e16.mean()) <= BF16_MEL_MEAN, (float(e16.max()), float(e16.mean()))
