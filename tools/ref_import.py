"""Import the REAL reference (`litfass`) in the build container and assemble a runnable
``FastSpeech2`` around its unmodified modules.  Used only by tools/gen_golden.py and by
tests that are skipped when /root/reference is absent (it never exists on the GPU box).

Nothing from the reference is copied: its package is imported from where it lies.  Packages it
imports that this image lacks (pytorch_lightning, torchaudio, wandb, pysdtw, numba, and the
dataset module's audio stack) are replaced by empty ``sys.modules`` stubs — none of them is
touched by ``FastSpeech2.forward``.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("FS2_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "litfass", "fastspeech2"))


def _install_stubs():
    import scipy.signal
    for n in ("cwt", "ricker"):  # removed in scipy>=1.15; only the off-path CWT transform uses them
        if not hasattr(scipy.signal, n):
            setattr(scipy.signal, n, None)

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []  # behave as a package for "import a.b"
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device("cpu")

        current_epoch = 0

    stub("pytorch_lightning", LightningModule=LightningModule)
    stub("torchaudio")
    stub("wandb")
    stub("pysdtw", SoftDTW=object)
    stub("numba", jit=lambda *a, **k: (lambda f: f))
    # litfass.dataset.datasets pulls pyworld/librosa/pandarallel/phones/srmrpy/seaborn/torchaudio
    if reference_available():
        sys.path.insert(0, REFERENCE_ROOT) if REFERENCE_ROOT not in sys.path else None
        import litfass.dataset  # real package
        stub("litfass.dataset.datasets", TTSDataset=object)
        stub("litfass.dataset.snr", SNR=object)


_loaded = None


def load_reference():
    """Returns (model_module, fastspeech2_module) of the real reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    _install_stubs()
    import litfass.fastspeech2.model as m
    try:
        import litfass.fastspeech2.fastspeech2 as fs
    except Exception as e:  # pragma: no cover - diagnostic
        raise RuntimeError(f"could not import litfass.fastspeech2.fastspeech2: {e!r}") from e
    _loaded = (m, fs)
    return _loaded


class Enc110(nn.Module):
    """torch-1.10 ``nn.TransformerEncoder`` semantics: plain loop, no final norm."""

    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def forward(self, src, src_key_padding_mask=None):
        x = src
        for layer in self.layers:
            x = layer(x, src_mask=None, src_key_padding_mask=src_key_padding_mask)
        return x


def build_reference_model(cfg, state_dict):
    """Assemble the reference ``FastSpeech2`` from its own modules exactly as
    fastspeech2.py:242-438 does, load ``state_dict`` (reference key names) and return it in eval
    mode.  ``forward`` is the unmodified reference method."""
    m, fs = load_reference()
    H = cfg.hidden
    obj = fs.FastSpeech2.__new__(fs.FastSpeech2)
    nn.Module.__init__(obj)
    obj.__dict__["hparams"] = SimpleNamespace(
        speaker_embedding_every_layer=False, prior_embedding_every_layer=False, priors=list(cfg.priors),
        variances=list(cfg.variances), fastdiff_variances=False,
        encoder_hidden=H, decoder_hidden=H, n_mels=cfg.n_mels, speaker_type="dvector",
    )
    obj.phone_embedding = nn.Embedding(cfg.n_phones, H, padding_idx=0)

    def layer(heads, F, k, dw):
        return m.ConformerEncoderLayer(H, heads, conv_in=H, conv_filter_size=F, conv_kernel=(k, 1),
                                       batch_first=True, dropout=0.1, conv_depthwise=dw)

    obj.encoder = Enc110([layer(cfg.encoder_head, cfg.encoder_conv_filter_size,
                                cfg.encoder_kernel_sizes[i], cfg.encoder_depthwise_conv)
                          for i in range(cfg.encoder_layers)])
    obj.positional_encoding = m.PositionalEncoding(H, dropout=0.1)
    nv = len(cfg.variances)
    obj.variance_adaptor = m.VarianceAdaptor(
        cfg.stats, list(cfg.variances), list(cfg.variance_levels[:nv]), list(cfg.variance_transforms[:nv]),
        list(cfg.variance_nlayers[:nv]), list(cfg.variance_kernel_size[:nv]), [0.5] * nv,
        cfg.variance_filter_size, cfg.variance_nbins, cfg.variance_depthwise_conv,
        cfg.duration_nlayers, False, cfg.duration_kernel_size, 0.5, cfg.duration_filter_size,
        cfg.duration_depthwise_conv, H, cfg.max_length * cfg.sampling_rate / cfg.hop_length)
    obj.decoder = Enc110([layer(cfg.decoder_head, cfg.decoder_conv_filter_size,
                                cfg.decoder_kernel_sizes[i], cfg.decoder_depthwise_conv)
                          for i in range(cfg.decoder_layers)])
    obj.linear = nn.Linear(H, cfg.n_mels)
    obj.speaker_embedding = m.SpeakerEmbedding(H, "dvector")
    obj.prior_embeddings = nn.ModuleDict({
        pr: m.PriorEmbedding(H, cfg.variance_nbins, cfg.stats[f"{pr}_prior"]) for pr in cfg.priors})
    # forward() calls self.fastdiff_linear unconditionally (fastspeech2.py:733) although it only
    # exists with a FastDiff vocoder attached; a dummy keeps the unmodified forward runnable and
    # does not touch `mel` (computed at :723).
    obj.fastdiff_linear = nn.Sequential(nn.Linear(H, H), nn.Linear(H, cfg.n_mels))
    obj.fastdiff_model = None
    obj.fastdiff_speaker_generator = None
    own = obj.state_dict()
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
    missing = [k for k in own if k not in sd and not k.startswith("fastdiff_linear")]
    extra = [k for k in sd if k not in own]
    if missing or extra:
        raise RuntimeError(f"state_dict naming drifted: missing={missing[:5]} extra={extra[:5]}")
    obj.load_state_dict(sd, strict=False)
    obj.eval()
    return obj


@torch.no_grad()
def run_reference(cfg, state_dict, phones, speaker, capture=True, priors=None, teacher_targets=None):
    """Run the unmodified reference forward; optionally capture intermediates with hooks."""
    model = build_reference_model(cfg, state_dict)
    inter = {}
    hooks = []
    if capture:
        hooks.append(model.encoder.register_forward_hook(lambda mod, i, o: inter.__setitem__("encoder_out", o)))
        hooks.append(model.decoder.register_forward_hook(lambda mod, i, o: inter.__setitem__("decoder_out", o)))
        hooks.append(model.variance_adaptor.register_forward_hook(
            lambda mod, i, o: inter.__setitem__("adaptor_out", o["x"])))
        def _lr_hook(mod, i, o):  # must return None or it would replace the module output
            inter.setdefault("regulated", o[0])
        hooks.append(model.variance_adaptor.length_regulator.register_forward_hook(_lr_hook))
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):  # "Zero duration, setting to 1" prints (model.py:309)
        batch = {"phones": torch.as_tensor(phones), "speaker": torch.as_tensor(speaker)}
        for k, v in (priors or {}).items():
            batch[k] = np.asarray(v)  # forward() wraps it with torch.tensor(...) (fastspeech2.py:690)
        if teacher_targets is not None:  # the Lightning hooks' teacher-forced call self(batch) (fastspeech2.py:787,800)
            for k, v in teacher_targets.items():
                batch[k] = torch.as_tensor(np.asarray(v))
            np.random.seed(0)  # tf_val = np.random.uniform(0, 1) <= 1.0 is always True (model.py:272)
            out = model(batch)
        else:
            out = model(batch, inference=True)
    for h in hooks:
        h.remove()
    out = {k: v for k, v in out.items() if k != "fastdiff_var"}
    out["_intermediates"] = inter
    out["_stdout"] = buf.getvalue()
    return out
