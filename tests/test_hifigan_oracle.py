"""CPU: the HiFi-GAN oracle against the fixtures captured from the reference's own Generator
(tools/gen_golden_hifigan.py), plus the host-side weight-norm folding of the mirror."""
import json
import os

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.hifigan import HifiGanConfig, fold_weight_norm, state_dict_spec, synth_state_dict
from oracle import hifigan_cpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["hifigan_two_stage", "hifigan_v1"]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = HifiGanConfig.from_json(str(z["config"]))
    return z, cfg, synth_state_dict(cfg, int(z["seed"]))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_generator(name):
    z, cfg, sd = load(name)
    wav = hifigan_cpu.synthesize(sd, cfg, torch.from_numpy(z["mel"]), torch.from_numpy(z["lengths"]))
    for b, n in enumerate(z["lengths"]):
        ref = torch.from_numpy(z[f"wav_{b}"])
        got = wav[b, :n * cfg.hop]
        assert ref.numel() == n * cfg.hop
        assert float((got - ref).abs().max()) <= 2e-5      # fp32 CPU vs fp32 CPU, different conv schedules
        assert float(wav[b, n * cfg.hop:].abs().max()) == 0.0 if n * cfg.hop < wav.shape[1] else True
        i16 = (got.numpy()[None] * 32768.0).astype("int16")  # Synthesiser.__call__ (__init__.py:39-43)
        assert int(np.abs(i16.astype(np.int32) - z[f"int16_{b}"].astype(np.int32)).max()) <= 1


@pytest.mark.parametrize("name", CASES)
def test_fold_weight_norm_recovers_plain_weights(name):
    z, cfg, sd = load(name)
    ck = {}
    for k, w in sd.items():
        if k.endswith(".weight"):
            t = torch.from_numpy(w)
            ck[k[:-7] + ".weight_g"] = t.flatten(1).norm(dim=1).reshape(-1, *([1] * (t.ndim - 1)))
            ck[k[:-7] + ".weight_v"] = t * 3.0   # any rescaling of v must fold away
        else:
            ck[k] = w
    # the reference-side checkpoint tensors the fixture kept pin the norm convention (dim 0 kept)
    assert np.allclose(ck["conv_pre.weight_g"].numpy(), z["ck_conv_pre_g"], rtol=1e-6)
    assert np.allclose(ck["ups.0.weight_g"].numpy(), z["ck_ups0_g"], rtol=1e-6)
    plain = fold_weight_norm(ck)
    assert set(plain) == set(state_dict_spec(cfg))
    for k in sd:
        assert plain[k].shape == tuple(state_dict_spec(cfg)[k])
        assert np.allclose(plain[k], sd[k], rtol=1e-5, atol=1e-7), k


def test_config_rejects_unsupported():
    with pytest.raises(ValueError):
        HifiGanConfig(resblock="2")
    with pytest.raises(ValueError):
        HifiGanConfig(upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[15, 16, 4, 4])
    with pytest.raises(ValueError):
        HifiGanConfig(upsample_initial_channel=64)   # last stage would have 4 channels
    assert HifiGanConfig().hop == 256 and HifiGanConfig().channels() == [512, 256, 128, 64, 32]
    assert json.loads(json.dumps(HifiGanConfig().__dict__))["num_mels"] == 80
