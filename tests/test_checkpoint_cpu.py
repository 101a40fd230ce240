"""Lightning-style checkpoints: a pickle written by torch.save of the dict Lightning writes - hyper_parameters
as pytorch_lightning's AttributeDict, state_dict of tensors, the on_save_checkpoint extras (fastspeech2.py:622-634) - read
back without Lightning installed, the class-default CWT pitch transform included, strict and tolerant loads."""
import sys
import types

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd import checkpoint as ck
from lightningfastspeech2_amd.config import Fs2Config
from lightningfastspeech2_amd.weights import state_dict_spec, synth_state_dict


def _cfg(**kw):
    base = dict(n_phones=30, encoder_hidden=32, decoder_hidden=32, encoder_head=2, decoder_head=2, encoder_layers=1,
                decoder_layers=1, encoder_kernel_sizes=[3], decoder_kernel_sizes=[3], encoder_conv_filter_size=64,
                decoder_conv_filter_size=64, variance_filter_size=32, variance_nlayers=[1, 1, 1], duration_filter_size=32,
                variance_nbins=8, n_mels=4, variance_transforms=["cwt", "none", "none"],
                stats={"pitch": {"min": 60.0, "max": 400.0, "mean": 180.0, "std": 40.0},
                       "energy": {"min": -3.0, "max": 3.0, "mean": 0.0, "std": 1.0},
                       "snr": {"min": -1.0, "max": 4.0, "mean": 1.2, "std": 2.0}})
    base.update(kw)
    return Fs2Config(**base)


def _write_lightning_checkpoint(path, cfg, sd, extra_state=None):
    """Pickle it the way Lightning does: hyper_parameters is an instance of
    pytorch_lightning.utilities.parsing.AttributeDict.  The class is provided by a throw-away module under that name and
    REMOVED again before loading, which is the situation of a machine without Lightning."""
    names = ["pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"]
    saved = {n: sys.modules.get(n) for n in names}
    mods = [types.ModuleType(n) for n in names]
    AD = type("AttributeDict", (dict,), {"__module__": names[2]})
    mods[2].AttributeDict = AD
    mods[0].utilities, mods[1].parsing = mods[1], mods[2]
    for n, m in zip(names, mods):
        sys.modules[n] = m
    try:
        hp = AD({k: v for k, v in cfg.to_dict().items() if k not in ("stats", "n_phones")})
        hp.update(lr=1e-4, warmup_steps=4000, fastdiff_variances=False, variance_dropout=[0.5, 0.5, 0.5])  # ignored extras
        ckpt = {"epoch": 3, "global_step": 1234, "pytorch-lightning_version": "1.5.10",
                "state_dict": {**{k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, **(extra_state or {})},
                "hyper_parameters": hp, "optimizer_states": [{"state": {}}], "lr_schedulers": [],
                "stats": cfg.stats, "phone2id": {f"p{i}": i for i in range(cfg.n_phones)},
                "speaker2dvector": {"spk0": np.zeros(256, np.float32)}, "speaker2priors": {"spk0": {"pitch": 1.0}}}
        torch.save(ckpt, path)
    finally:
        for n in names:
            if saved[n] is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = saved[n]


def test_reads_a_lightning_pickle_without_lightning(tmp_path):
    cfg = _cfg()
    sd = synth_state_dict(cfg, 3)
    path = tmp_path / "lit_model.ckpt"
    _write_lightning_checkpoint(path, cfg, sd, {"fastdiff_linear.0.weight": torch.zeros(32, 32)})
    assert "pytorch_lightning" not in sys.modules
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)  # the stock loader cannot resolve AttributeDict
    c = ck.read_checkpoint(path)
    assert isinstance(c["hyper_parameters"], ck.AttributeDict) and c["hyper_parameters"].encoder_hidden == 32
    cfg2 = ck.config_from_checkpoint(c)
    assert cfg2.to_dict() == cfg.to_dict() and cfg2.is_cwt(0)  # the class default transform [cwt, none, none] survives
    w, rep = ck.resolve_state_dict(cfg2, c["state_dict"])
    assert set(w) == set(state_dict_spec(cfg)) and rep["dropped"] == ["fastdiff_linear.0.weight"]
    assert w["variance_adaptor.encoders.pitch.predictor.linear.weight"].shape == (10, 32)
    assert w["variance_adaptor.encoders.pitch.mean_std_linear.weight"].shape == (2, 32)
    for k in sd:
        np.testing.assert_array_equal(w[k], sd[k])


def test_strict_and_tolerant_loads(tmp_path, capsys):
    cfg = _cfg()
    sd = dict(synth_state_dict(cfg, 3))
    bad = dict(sd)
    bad["linear.weight"] = np.zeros((5, 32), np.float32)          # e.g. trained with another n_mels
    del bad["speaker_embedding.projection.bias"]
    with pytest.raises(ValueError, match="linear.weight"):
        ck.resolve_state_dict(cfg, bad)
    w, rep = ck.resolve_state_dict(cfg, bad, tolerant=True, init_seed=9)
    out = capsys.readouterr().out
    assert "Skip loading parameter: linear.weight, required shape: (4, 32), loaded shape: (5, 32)" in out  # fastspeech2.py:605-609
    assert rep == {"skipped": ["linear.weight"], "missing": ["speaker_embedding.projection.bias"], "dropped": []}
    fresh = synth_state_dict(cfg, 9)
    np.testing.assert_array_equal(w["linear.weight"], fresh["linear.weight"])  # keeps a fresh initialisation, as the reference
    np.testing.assert_array_equal(w["linear.bias"], sd["linear.bias"])
