"""Lightning checkpoint -> (config, weights, extras) for the mel forward.

What ``FastSpeech2.load_from_checkpoint`` + ``on_load_checkpoint`` do in the reference
(litfass/generate.py:106-112, litfass/fastspeech2/fastspeech2.py:530-620), as host code with no Lightning
dependency: read the pickle, rebuild the shape-deciding hparams, pick up the extras the callers read
(``stats``, ``phone2id``, ``speaker2dvector``, ...) and hand back the state_dict entries the path needs.

A checkpoint written by Lightning pickles its ``hyper_parameters`` as
``pytorch_lightning.utilities.parsing.AttributeDict`` (a dict subclass).  On a machine without Lightning the class
cannot be imported, so unpickling resolves any class of the ``pytorch_lightning`` / ``lightning`` packages it cannot
import to a plain attribute-dict stand-in: the data is all this path needs.
"""
from __future__ import annotations

import importlib
import io
import pickle
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .config import Fs2Config
from .weights import state_dict_spec, synth_state_dict


class AttributeDict(dict):
    """dict with attribute access: the stand-in for Lightning's class of the same name."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if module.split(".")[0] in ("pytorch_lightning", "lightning", "lightning_fabric"):
                return AttributeDict
            raise


class _PickleModule:
    """What torch.load wants as ``pickle_module``."""
    __name__ = "lightningfastspeech2_amd.checkpoint._PickleModule"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _Unpickler(io.BytesIO(b), **kw).load())
    dump, dumps = pickle.dump, pickle.dumps
    PickleError, UnpicklingError, PicklingError = pickle.PickleError, pickle.UnpicklingError, pickle.PicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def read_checkpoint(path) -> dict:
    """torch.load of a ``lit_model.ckpt`` onto the CPU (needs the pickle loader: a Lightning checkpoint is more than
    tensors), tolerant of a missing Lightning installation."""
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)


def config_from_checkpoint(checkpoint: dict) -> Fs2Config:
    hp = checkpoint["hyper_parameters"]
    return Fs2Config.from_hparams(hp, stats=checkpoint["stats"], n_phones=len(checkpoint["phone2id"]))


def resolve_state_dict(cfg: Fs2Config, state_dict: Dict[str, object], *, tolerant: bool = False, init_seed: int = 0,
                       log=print) -> Tuple[Dict[str, np.ndarray], Dict[str, list]]:
    """The tensors of the forward path, by the reference's key names.

    strict (default): a tensor of the architecture that is missing or has another shape is an error naming it.
    ``tolerant=True`` is the reference's behaviour (fastspeech2.py:598-620 + ``strict=False`` in train.py:240-250): a
    shape-mismatched entry is skipped with the reference's own message and that parameter keeps a fresh default
    initialisation (``synth_state_dict(cfg, init_seed)``: the same distributions torch's initialisers draw from), a
    missing one likewise, and entries the model does not have are dropped with a message.
    Returns (weights, report) with report = {"skipped": [...], "missing": [...], "dropped": [...]}."""
    spec = state_dict_spec(cfg)
    out, report = {}, {"skipped": [], "missing": [], "dropped": []}
    fresh = None
    for name, shape in spec.items():
        v = state_dict.get(name)
        if v is not None:
            a = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)
            if tuple(a.shape) == tuple(shape):
                out[name] = np.ascontiguousarray(a, dtype=np.float32)
                continue
            if not tolerant:
                raise ValueError(f"checkpoint tensor '{name}' has shape {tuple(a.shape)}, the architecture needs {tuple(shape)}")
            log(f"Skip loading parameter: {name}, required shape: {tuple(shape)}, loaded shape: {tuple(a.shape)}")
            report["skipped"].append(name)
        else:
            if not tolerant:
                raise KeyError(f"checkpoint state_dict is missing '{name}'")
            report["missing"].append(name)
        if fresh is None:
            fresh = synth_state_dict(cfg, init_seed)
        out[name] = fresh[name]
    for name in state_dict:
        if name not in spec:
            report["dropped"].append(name)
            if tolerant:
                log(f"Dropping parameter {name}")
    return out, report


EXTRA_KEYS = ("speaker2id", "speaker2dvector", "speaker2priors", "speaker_gmms", "dvector_gmms")  # fastspeech2.py:571-587
