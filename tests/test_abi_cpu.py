"""CPU-side checks of the C-ABI boundary (no compute calls: there is no GPU here)."""
import ctypes as C
import os

import numpy as np
import pytest

from lightningfastspeech2_amd import _lib
from lightningfastspeech2_amd.config import Fs2Config, preset
from lightningfastspeech2_amd.weights import state_dict_spec, synth_state_dict


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    names = _lib.declared_symbols()
    assert len(names) >= 20 and "fs2_encode" in names and "fs2_decode" in names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.fs2_abi_version() == _lib.FS2_ABI_VERSION


def test_struct_layout_matches_header():
    # sizes follow from include/fs2.h: 8 + (4+32)*2 ints, 1 int, 4*32 chars, 2*4 ints, 2*4 floats, 7 ints, n_priors,
    # FS2_MAX_PRIORS names (ABI v3: the shipped recipe lists five priors, scripts/train.sh:49), var_cwt
    assert _lib.FS2_ABI_VERSION == 4 and _lib.FS2_MAX_PRIORS == 8  # v4: var_level (phone-level variances)
    assert C.sizeof(_lib.Fs2ConfigC) == 4 * (8 + 36 + 36 + 1) + 4 * 32 + 4 * (4 + 4) + 4 * (4 + 4) + 4 * 7 + 4 + 8 * 32 + 4 * 4 + 4 * 4  # ... var_cwt, var_level
    assert C.sizeof(_lib.Fs2OutputsC) == 8 * (5 + 3 * _lib.FS2_MAX_VARIANCES)


def _create(lib, cfg, dtype=_lib.FS2_F32):
    cc = _lib.config_to_c(cfg, dtype)
    h = C.c_void_p()
    st = lib.fs2_create(C.byref(cc), C.byref(h))
    return st, h


def test_create_validates_config(lib):
    st, h = _create(lib, preset("c2"), _lib.FS2_BF16)
    assert st == 0
    lib.fs2_destroy(h)
    bad = Fs2Config(encoder_hidden=96, decoder_hidden=96, encoder_conv_filter_size=192,
                    decoder_conv_filter_size=192, variance_filter_size=96, duration_filter_size=96,
                    encoder_head=1, decoder_head=1)
    st, h = _create(lib, bad)
    assert st == 2 and b"multiple of 64" in lib.fs2_last_error(h)  # FS2_ERR_SHAPE, loud and specific
    lib.fs2_destroy(h)
    st, h = _create(lib, Fs2Config(encoder_head=16, decoder_head=16))  # head dim 16 unsupported
    assert st == 2
    lib.fs2_destroy(h)


def test_recipe_with_five_priors_fits_the_abi(lib):
    # scripts/train.sh:27,49 of the reference: four variances, five priors
    names = ["pitch", "energy", "snr", "srmr"]
    priors = ["energy", "duration", "snr", "pitch", "srmr"]
    stats = {v: {"min": -3.0, "max": 3.0, "mean": 0.0, "std": 1.0} for v in names}
    stats.update({f"{p}_prior": {"min": -2.0, "max": 2.0, "mean": 0.0, "std": 1.0} for p in priors})
    cfg = Fs2Config(variances=names, variance_levels=["frame"] * 4, variance_transforms=["none"] * 4,
                    variance_nlayers=[5] * 4, variance_kernel_size=[3] * 4, priors=priors, stats=stats,
                    decoder_layers=6, decoder_kernel_sizes=[9] * 6, duration_nlayers=5,
                    encoder_depthwise_conv=False, decoder_depthwise_conv=False)
    st, h = _create(lib, cfg, _lib.FS2_BF16)
    assert st == 0, lib.fs2_last_error(h)
    lib.fs2_destroy(h)
    cfg.priors = [f"p{i}" for i in range(9)]
    with pytest.raises(ValueError, match="too many priors"):
        _lib.config_to_c(cfg, _lib.FS2_BF16)


def test_load_weight_checks_names_and_shapes(lib):
    cfg = preset("ref-default")
    st, h = _create(lib, cfg)
    assert st == 0
    sd = synth_state_dict(cfg, 0)

    def load(name, arr):
        a = np.ascontiguousarray(arr, np.float32)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        return lib.fs2_load_weight(h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)

    # every tensor of the reference state_dict (reference key names) is accepted ...
    for name, shape in state_dict_spec(cfg).items():
        assert load(name, sd[name]) == 0, name
    # ... off-path / misspelled names and wrong shapes are refused
    assert load("fastdiff_linear.0.weight", np.zeros((256, 256))) == 4
    assert load("encoder.layers.0.conv1.weight", np.zeros((1024, 256, 5))) == 4  # dense key on a depth-wise config
    assert load("linear.weight", np.zeros((80, 255))) == 4
    assert b"shape mismatch" in lib.fs2_last_error(h)
    # decode before encode is a state error, not a crash
    out = _lib.Fs2OutputsC()
    assert lib.fs2_decode(h, C.byref(out), None) == 5
    lib.fs2_destroy(h)


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lightningfastspeech2_amd.model import FastSpeech2
    cfg = preset("ref-default")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FastSpeech2(cfg, synth_state_dict(cfg, 0))


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "lightningfastspeech2_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle_cpu" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_split_k_choice_is_a_pure_shape_function(lib):
    """fs2_op_gemm_splitk_choice runs on the host: the shapes the training step meets.  A long reduction over few row tiles
    splits (the encoder-side data-gradient convs), launches that already fill the chip or reduce over a short K do not."""
    BF16, F32 = 1, 0
    f = lib.fs2_op_gemm_splitk_choice
    assert f(BF16, 8192, 256, 1024, 9, 256) == 4      # C2 encoder conv1 data gradient: 64 row tiles x 4 = one round of 256 CUs
    assert f(BF16, 2048, 1024, 4096, 9, 256) == 4     # C5, 8 utterances per GPU
    assert f(BF16, 49152, 256, 1024, 9, 1536) == 1    # decoder: 384 row tiles already
    assert f(BF16, 8192, 1024, 256, 1, 8192) == 1     # short reduction
    assert f(BF16, 8192, 128, 1024, 9, 256) == 1      # narrower than the slab kernel's column tile
    assert f(BF16, 8192, 256, 1024, 2, 256) == 1      # even tap count: not a slab-kernel conv
    for dt in (F32, BF16):
        for args in ((8192, 256, 1024, 9, 256), (2048, 1024, 4096, 9, 256), (300, 256, 2048, 1, 300)):
            k = f(dt, *args)
            assert k >= 1 and (args[2] // (64 if dt == BF16 else 32)) % k == 0


def test_knob_setter_rejects_undefined_values(lib):
    """fs2_op_set_gemm_variant takes closed ranges - a typo in FS2_GEMM_KNOBS is an error, not a silent default; the
    switches removed in r05 (measured-slower forms: 1211, 211, 1301 / 1302, 301, 311 and their "off" values) are undefined now."""
    FS2_OK, FS2_ERR_ARG = 0, lib.fs2_op_set_gemm_variant(-1)
    assert FS2_ERR_ARG != FS2_OK
    for bad in (1300, 1301, 1302, 1319, 1322, 1399, 1403, 1205, 1210, 1211, 1299, 1102, 1002, 910, 802, 702, 502, 310, 311, 300, 301, 210, 211, 222, 203, 250, 251, 252, 253, 8, 100000):
        assert lib.fs2_op_set_gemm_variant(bad) == FS2_ERR_ARG, bad
    # the defaults (each is a defined value) leave the process as it was
    for ok in (0, 202, 201, 221, 230, 231, 500, 701, 801, 909, 904, 1001, 1100, 1203, 1321, 1340, 1341, 1401, 1501):
        assert lib.fs2_op_set_gemm_variant(ok) == FS2_OK, ok


def test_library_has_no_mutable_tuning_globals():
    """SURVEY 8b "no global state (one handle per device)" / VERDICT r04 item 6: the A/B switches live in a per-engine / per-thread
    `fs2::Tuning`, not in process-wide `int g_*` variables every engine and pipeline thread reads unsynchronised.  By `nm`: no
    defined data symbol named fs2::g_* except the read-only device zero page."""
    import subprocess
    from lightningfastspeech2_amd import _lib
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    bad = []
    for line in out.splitlines():
        parts = line.split(None, 2)
        if len(parts) == 3 and parts[1] in "bBdD" and "fs2::g_" in parts[2] and "(" not in parts[2] and "g_zero_page" not in parts[2]:
            bad.append(parts[2])
    assert not bad, bad
