"""GPU parity of the HiFi-GAN generator (fs2_voc_* C ABI) against the reference fixtures and the
CPU oracle.  Tolerances: fp32 mode max|d wav| <= 1e-3 (samples are in [-1, 1]; the same bar the mel
forward uses); bf16 mode is reported against a looser, stated bound."""
import os

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.hifigan import HifiGan, HifiGanConfig, Synthesiser, synth_state_dict
from oracle import hifigan_cpu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
F32_TOL = 1e-3
BF16_TOL = 6e-2   # ~70 bf16-rounded conv layers deep; stated, not claimed as parity


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = HifiGanConfig.from_json(str(z["config"]))
    return z, cfg, synth_state_dict(cfg, int(z["seed"]))


@pytest.mark.parametrize("name", ["hifigan_two_stage", "hifigan_v1"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_generator_matches_reference_fixture(name, precision):
    z, cfg, sd = load(name)
    g = HifiGan(cfg, sd, precision=precision)
    mel, lengths = torch.from_numpy(z["mel"]), torch.from_numpy(z["lengths"])
    wav = g.synthesize(mel, lengths).cpu()
    tol = F32_TOL if precision == "fp32" else BF16_TOL
    for b, n in enumerate(z["lengths"]):
        ref = torch.from_numpy(z[f"wav_{b}"])
        err = float((wav[b, :n * cfg.hop] - ref).abs().max())
        assert err <= tol, (name, precision, b, err)
        assert float(wav[b, n * cfg.hop:].abs().sum()) == 0.0
    # every utterance alone == the same utterance inside the ragged batch (no cross-utterance leakage)
    for b, n in enumerate(z["lengths"]):
        alone = g.synthesize(mel[b:b + 1, :n]).cpu()
        assert torch.equal(alone[0], wav[b, :n * cfg.hop])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_stage_outputs_and_tile_seams(precision):
    """T = 45 frames spans several workgroup tiles in every stage (224 rows = 28 frames at 256
    channels ... 1792 rows = 7 frames at 32); compare every stage output and the samples with the
    oracle, ragged lengths included."""
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 5)
    rs = np.random.RandomState(9)
    mel = torch.from_numpy((rs.standard_normal((3, 45, 80)) * 1.5 - 4.0).astype(np.float32))
    lengths = torch.tensor([45, 29, 1], dtype=torch.int32)
    ref_wav, ref_stages = hifigan_cpu.synthesize(sd, cfg, mel, lengths, return_stages=True)
    g = HifiGan(cfg, sd, precision=precision)
    wav = g.synthesize(mel, lengths).cpu()
    rel = 1e-4 if precision == "fp32" else 4e-2
    for s in range(len(cfg.upsample_rates) + 1):
        got = g.debug_stage(s).cpu()
        up = int(np.prod(cfg.upsample_rates[:s])) if s else 1
        for b, n in enumerate(lengths.tolist()):
            ref = ref_stages[b][s]
            d = float((got[b, :n * up] - ref).abs().max())
            assert d <= rel * (float(ref.abs().max()) + 1.0), (s, b, d)
    tol = F32_TOL if precision == "fp32" else BF16_TOL
    assert float((wav - ref_wav).abs().max()) <= tol


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_resblock_equals_conv_by_conv(precision):
    """The LDS-resident resblock kernel (32/64-channel stages) against the same generator run conv by
    conv: same arithmetic up to the storage rounding of the residual stream, several tiles, ragged."""
    from lightningfastspeech2_amd import _lib
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 8)
    rs = np.random.RandomState(2)
    mel = torch.from_numpy((rs.standard_normal((2, 23, 80)) * 1.5 - 4.0).astype(np.float32))
    lengths = torch.tensor([23, 10], dtype=torch.int32)
    g = HifiGan(cfg, sd, precision=precision)
    try:
        _lib.load().fs2_op_set_vocoder_fused_resblock(0)
        plain = g.synthesize(mel, lengths).cpu()
        s_plain = [g.debug_stage(s).cpu() for s in (3, 4)]
        _lib.load().fs2_op_set_vocoder_fused_resblock(1)
        fused = g.synthesize(mel, lengths).cpu()
        s_fused = [g.debug_stage(s).cpu() for s in (3, 4)]
    finally:
        _lib.load().fs2_op_set_vocoder_fused_resblock(1)
    tol = 2e-5 if precision == "fp32" else 5e-2
    for a, b, up in zip(s_plain, s_fused, (128, 256)):
        for u, n in enumerate(lengths.tolist()):
            ref = a[u, :n * up]
            assert float((b[u, :n * up] - ref).abs().max()) <= tol * (float(ref.abs().max()) + 1.0)
    assert float((plain - fused).abs().max()) <= (2e-5 if precision == "fp32" else BF16_TOL)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_activated_stream_equals_raw_stream(precision):
    """Between LDS-resident launches the residual stream (and a fully resident stage's upsampler output) is stored
    as lrelu(x) and the consumers fill their slabs by LDS-DMA (knob 1); knob 9 stores raw x everywhere.  Same
    arithmetic up to one storage rounding per hop; ragged batch, several tiles per utterance."""
    from lightningfastspeech2_amd import _lib
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 11)
    rs = np.random.RandomState(4)
    mel = torch.from_numpy((rs.standard_normal((3, 37, 80)) * 1.5 - 4.0).astype(np.float32))
    lengths = torch.tensor([37, 12, 1], dtype=torch.int32)
    g = HifiGan(cfg, sd, precision=precision)
    try:
        _lib.load().fs2_op_set_vocoder_fused_resblock(9)
        raw = g.synthesize(mel, lengths).cpu()
        s_raw = [g.debug_stage(s).cpu() for s in (2, 3, 4)]
        _lib.load().fs2_op_set_vocoder_fused_resblock(1)
        act = g.synthesize(mel, lengths).cpu()
        s_act = [g.debug_stage(s).cpu() for s in (2, 3, 4)]
    finally:
        _lib.load().fs2_op_set_vocoder_fused_resblock(1)
    tol = 2e-5 if precision == "fp32" else 5e-2
    for a, b, up in zip(s_raw, s_act, (64, 128, 256)):
        for u, n in enumerate(lengths.tolist()):
            ref = a[u, :n * up]
            assert float((b[u, :n * up] - ref).abs().max()) <= tol * (float(ref.abs().max()) + 1.0)
    assert float((raw - act).abs().max()) <= (2e-5 if precision == "fp32" else BF16_TOL)


def test_synthesiser_mirror_int16():
    """Synthesiser(mel) -> int16 (1, T*256), the reference wrapper's contract (__init__.py:37-43),
    from a checkpoint-form state_dict ({"generator": weight_g / weight_v ...})."""
    z, cfg, sd = load("hifigan_v1")
    ck = {}
    for k, w in sd.items():
        if k.endswith(".weight"):
            t = torch.from_numpy(w)
            ck[k[:-7] + ".weight_g"] = t.flatten(1).norm(dim=1).reshape(-1, *([1] * (t.ndim - 1)))
            ck[k[:-7] + ".weight_v"] = t.clone()
        else:
            ck[k] = torch.from_numpy(w)
    synth = Synthesiser(device="cuda:0", checkpoint={"generator": ck}, precision="fp32")
    n = int(z["lengths"][0])
    out = synth(torch.from_numpy(z["mel"][0, :n]))
    assert out.dtype == np.int16 and out.shape == (1, n * 256)
    assert int(np.abs(out.astype(np.int32) - z["int16_0"].astype(np.int32)).max()) <= 40   # 1e-3 * 32768
    with pytest.raises(FileNotFoundError):
        Synthesiser(device="cuda:0")


def test_errors():
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 1)
    bad = dict(sd)
    del bad["resblocks.7.convs2.1.bias"]
    with pytest.raises(KeyError):
        HifiGan(cfg, bad)
    g = HifiGan(cfg, sd)
    with pytest.raises(ValueError):
        g.synthesize(torch.zeros(1, 4, 64))


def test_mel_forward_into_vocoder_matches_per_utterance_reference_flow():
    """SpeechGenerator.generate_samples (generator.py:151-223): FastSpeech2 forward -> per-utterance
    unpadded mel -> generator -> int16 -> float32/32767.  HIP: the padded mel batch stays in HBM and
    the generator takes the valid frame counts; oracle: the reference's own per-utterance loop."""
    from lightningfastspeech2_amd.config import Fs2Config
    from lightningfastspeech2_amd.model import FastSpeech2
    from lightningfastspeech2_amd.synthesis import SpeechGenerator
    from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict as fs2_sd
    from oracle import oracle_cpu
    cfg = Fs2Config(n_phones=40, encoder_hidden=64, decoder_hidden=64, encoder_head=2, decoder_head=2,
                    encoder_layers=2, decoder_layers=2, encoder_kernel_sizes=[3, 5], decoder_kernel_sizes=[5, 3],
                    encoder_conv_filter_size=128, decoder_conv_filter_size=128, encoder_depthwise_conv=False,
                    decoder_depthwise_conv=False, variance_filter_size=64, variance_depthwise_conv=False,
                    variance_nlayers=[2, 2, 2], duration_filter_size=64, duration_depthwise_conv=False, n_mels=80)
    sd = fs2_sd(cfg, 3, randomize_norm=True, duration_bias=1.3)
    inp = synth_inputs(cfg, 3, 12, seed=5, lengths=[12, 8, 3])
    vcfg = HifiGanConfig(upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=128,
                         resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2, 3], [1, 3, 5]])
    vsd = synth_state_dict(vcfg, 4)
    ref = oracle_cpu.forward(sd, cfg, inp["phones"], inp["speaker"])
    model = FastSpeech2(cfg, sd, precision="fp32", device="cuda:0")
    gen = SpeechGenerator(model, HifiGan(vcfg, vsd, precision="fp32"))
    out = gen.generate_samples({"phones": torch.from_numpy(inp["phones"]), "speaker": torch.from_numpy(inp["speaker"])},
                               return_duration=True)
    assert out["fs"] == 22050 and len(out["audios"]) == 3
    for b in range(3):
        keep = ~ref["tgt_mask"][b]
        n = int(keep.sum())
        wav = hifigan_cpu.synthesize(vsd, vcfg, ref["mel"][b][keep].unsqueeze(0))[0]
        want = (wav.numpy() * 32768.0).astype("int16").astype(np.float32) / 32767.0
        got = out["audios"][b]
        assert got.dtype == np.float32 and got.shape == (n * vcfg.hop,)
        assert float(np.abs(got - want).max()) <= F32_TOL
        assert torch.equal(out["durations"][b], ref["duration_rounded"][b])


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_full_size_time_shift_equivariance(precision):
    """Size-independent property at the benchmark's utterance length (1536 frames -> 393 216 samples, far
    beyond what the CPU oracle runs in seconds): the generator is a stack of convolutions, so dropping the
    first k frames of the mel shifts the waveform by k*256 samples away from the utterance edges.  Every
    sample keeps its own arithmetic order whatever tile it falls in, hence the comparison is bit-exact;
    a tile-seam, halo or phase-ordering error in any of the 40 launches breaks it."""
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 21)
    g = HifiGan(cfg, sd, precision=precision)
    rs = np.random.RandomState(4)
    T, k, margin = (1536, 8, 32) if precision == "bf16" else (384, 5, 32)
    mel = torch.from_numpy((rs.standard_normal((2, T, 80)) * 1.5 - 4.0).astype(np.float32))
    full = g.synthesize(mel).cpu()
    shifted = g.synthesize(mel[:, k:].contiguous()).cpu()
    hop = cfg.hop
    a = full[:, (k + margin) * hop:(T - margin) * hop]
    b = shifted[:, margin * hop:(T - k - margin) * hop]
    assert torch.isfinite(full).all() and float(full.abs().max()) <= 1.0
    assert torch.equal(a, b)
    assert not torch.equal(full[:, :margin * hop // 2], shifted[:, :margin * hop // 2])   # edges do differ


@pytest.mark.parametrize("rates,kernels", [((4, 2), (8, 4)), ((4, 2), (12, 6)), ((2, 2), (2, 2)), ((8, 2), (16, 6))])
def test_transposed_conv_tap_windows(rates, kernels):
    """ConvTranspose1d(k, stride s, padding (k - s) / 2) for k = 2s (every phase has two of the three taps: the engine runs
    two tap windows, `shift_from`), k = 3s (all three taps everywhere: the plain three-tap form) and k = s (one tap), fp32 against
    the oracle on ragged lengths that cross several tiles (reference: third_party/hifigan/models.py:131-136, 149-150)."""
    cfg = HifiGanConfig(upsample_rates=list(rates), upsample_kernel_sizes=list(kernels), upsample_initial_channel=128,
                        resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]])
    sd = synth_state_dict(cfg, 11)
    rs = np.random.RandomState(3)
    mel = torch.from_numpy((rs.standard_normal((3, 150, 80)) * 1.5 - 4.0).astype(np.float32))
    lengths = torch.tensor([150, 97, 2], dtype=torch.int32)
    ref_wav, ref_stages = hifigan_cpu.synthesize(sd, cfg, mel, lengths, return_stages=True)
    g = HifiGan(cfg, sd, precision="fp32")
    wav = g.synthesize(mel, lengths).cpu()
    for s in range(len(rates) + 1):
        got = g.debug_stage(s).cpu()
        up = int(np.prod(rates[:s])) if s else 1
        for b, n in enumerate(lengths.tolist()):
            ref = ref_stages[b][s]
            d = float((got[b, :n * up] - ref).abs().max())
            assert d <= 1e-4 * (float(ref.abs().max()) + 1.0), (rates, kernels, s, b, d)
    assert float((wav - ref_wav).abs().max()) <= F32_TOL


def test_wide_generator_takes_the_runtime_stride_conv_build():
    """1024 initial channels: the first transposed conv reads 1024-channel rows, a width outside the kernel's compile-time set
    (32 ... 512), so its K loop runs the runtime-stride build - fp32 against the oracle, ragged lengths."""
    cfg = HifiGanConfig(upsample_rates=[2, 2, 2], upsample_kernel_sizes=[4, 4, 4], upsample_initial_channel=1024,
                        resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 3, 5]])
    sd = synth_state_dict(cfg, 4)
    rs = np.random.RandomState(1)
    mel = torch.from_numpy((rs.standard_normal((2, 40, 80)) * 1.5 - 4.0).astype(np.float32))
    lengths = torch.tensor([40, 23], dtype=torch.int32)
    ref = hifigan_cpu.synthesize(sd, cfg, mel, lengths)
    ref = ref[0] if isinstance(ref, tuple) else ref
    wav = HifiGan(cfg, sd, precision="fp32").synthesize(mel, lengths).cpu()
    assert float((wav - ref).abs().max()) <= F32_TOL
