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


class _Opaque:
    """Stand-in for any OTHER Lightning class a checkpoint references (enum members rebuilt as ``cls("fit")``, callback /
    loop / progress state objects of some Lightning versions): accepts whatever the pickle hands it - constructor
    arguments, ``__setstate__`` of any shape, item / attribute assignment - and keeps it for inspection.  Nothing on the
    forward path reads these objects."""

    def __init__(self, *args, **kwargs):
        self.__dict__["_args"], self.__dict__["_kwargs"], self.__dict__["_state"] = args, kwargs, None

    def __setstate__(self, state):
        self.__dict__["_state"] = state
        if isinstance(state, dict):
            self.__dict__.update({k: v for k, v in state.items() if isinstance(k, str)})

    def __setitem__(self, k, v):
        self.__dict__.setdefault("_items", {})[k] = v

    def __call__(self, *args, **kwargs):  # an enum CLASS stand-in called with a value
        return _Opaque(*args, **kwargs)

    def append(self, v):
        self.__dict__.setdefault("_list", []).append(v)

    def extend(self, vs):
        self.__dict__.setdefault("_list", []).extend(vs)

    def __repr__(self):
        return f"<opaque Lightning object args={self._args!r}>"


_OPAQUE_CLASSES: Dict[Tuple[str, str], type] = {}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if module.split(".")[0] in ("pytorch_lightning", "lightning", "lightning_fabric", "lightning_lite"):
                if name == "AttributeDict":
                    return AttributeDict
                key = (module, name)  # one class per pickled name, so that NEWOBJ / REDUCE / BUILD all find a real type
                if key not in _OPAQUE_CLASSES:
                    _OPAQUE_CLASSES[key] = type(name, (_Opaque,), {"__module__": __name__, "_pickled_as": f"{module}.{name}"})
                return _OPAQUE_CLASSES[key]
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


# ---- optimizer state <-> Lightning's checkpoint layout -------------------------------------------------------------------
def parameter_order(cfg: Fs2Config) -> list:
    """Names of the trainable tensors in the order ``FastSpeech2.parameters()`` yields them in the reference - what
    ``torch.optim.AdamW(self.parameters())`` (fastspeech2.py:1166-1173) numbers 0..n-1 in ``optimizer.state_dict()``.
    Module registration order in ``FastSpeech2.__init__``: phone_embedding (fastspeech2.py:243), encoder (:249),
    positional_encoding (:296, a buffer only), variance_adaptor (:301-341), decoder (:347), linear (:385),
    prior_embeddings (:417-424), speaker_embedding (:428); inside a module, ``state_dict`` order minus the one buffer
    (``pe``).  The bucket edges ``bins`` of VarianceEncoder / PriorEmbedding ARE ``nn.Parameter(requires_grad=False)``
    (model.py:150,397): they take a slot in the optimizer's numbering and never get any state.  (FastDiff modules, when
    configured, would follow ``linear``; they are off this path.)"""
    names = [n for n in state_dict_spec(cfg) if n != "positional_encoding.pe"]
    top = ["phone_embedding", "encoder", "variance_adaptor", "decoder", "linear", "prior_embeddings", "speaker_embedding"]
    rank = {t: i for i, t in enumerate(top)}
    unknown = [n for n in names if n.split(".")[0] not in rank]
    if unknown:
        raise ValueError(f"parameter_order: no place for {unknown[:3]}")

    def key(n):
        # a module's own parameters come before its children's: VarianceEncoder = bins, predictor (registered first,
        # model.py:391-393), embedding, mean_std_linear (model.py:397-404); state_dict_spec lists the embedding before the predictor
        parts = n.split(".")
        sub = 0
        if parts[0] == "variance_adaptor" and parts[1] == "encoders":
            sub = {"bins": 0, "predictor": 1, "embedding": 2, "mean_std_linear": 3}[parts[3]]
        return (rank[parts[0]], sub if parts[0] == "variance_adaptor" and parts[1] == "encoders" else 0)

    out, i = [], 0
    # stable sort inside one variance encoder only: everything else keeps the spec's (= the modules') order
    while i < len(names):
        n = names[i]
        parts = n.split(".")
        if parts[0] == "variance_adaptor" and parts[1] == "encoders":
            j = i
            while j < len(names) and names[j].split(".")[:3] == parts[:3]:
                j += 1
            out += sorted(names[i:j], key=key)
            i = j
        else:
            out.append(n)
            i += 1
    return sorted(out, key=lambda n: rank[n.split(".")[0]])  # stable: keeps the in-module order


def to_lightning_optimizer_state(cfg: Fs2Config, opt_state: dict, *, lr: float, warmup_steps: int, betas=(0.9, 0.98), eps=1e-8,
                                 weight_decay=0.01) -> dict:
    """``Trainer.optimizer_state()`` -> the two checkpoint entries Lightning writes for the reference's optimizer and scheduler:
    ``{"optimizer_states": [AdamW.state_dict()], "lr_schedulers": [NoamLR.state_dict()]}`` - parameters numbered in
    ``parameter_order``, tensors in the reference's shapes, ``step`` per parameter as torch >= 1.12 keeps it (a 0-d fp32
    tensor; torch 1.10's int is accepted on the way back).  The current rate is NoamLR's for ``last_epoch`` = steps taken
    (noam.py:19-25)."""
    names = parameter_order(cfg)
    step = int(opt_state["step"])
    state = {}
    for i, n in enumerate(names):
        if step == 0 or n.endswith(".bins"):
            continue  # AdamW holds no state for a parameter before its first step, nor ever for the frozen bucket edges
        state[i] = {"step": torch.tensor(float(step)), "exp_avg": torch.as_tensor(opt_state["exp_avg"][n]).detach().cpu().clone(),
                    "exp_avg_sq": torch.as_tensor(opt_state["exp_avg_sq"][n]).detach().cpu().clone()}
    e = max(1, step)
    cur = lr * warmup_steps ** 0.5 * min(e ** -0.5, e * warmup_steps ** -1.5)
    group = {"lr": cur, "betas": list(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": lr,
             "params": list(range(len(names)))}
    sched = {"warmup_steps": warmup_steps, "base_lrs": [lr], "last_epoch": step, "_step_count": step + 1, "verbose": False,
             "_get_lr_called_within_step": False, "_last_lr": [cur]}
    return {"optimizer_states": [{"state": state, "param_groups": [group]}], "lr_schedulers": [sched]}


def from_lightning_optimizer_state(cfg: Fs2Config, checkpoint: dict, accumulate_grad_batches: int = 1) -> dict:
    """The inverse: a reference checkpoint's ``optimizer_states[0]`` (+ ``lr_schedulers[0]`` / ``global_step`` for the
    step count) -> what ``Trainer.load_optimizer_state`` takes.  Shapes are checked against the architecture; a checkpoint
    whose optimizer covers other parameters (FastDiff attached) is refused by count.  ``micro_step`` (the dropout-mask
    counter) resumes at step x ``accumulate_grad_batches`` unless the checkpoint carries the extra key ``fs2_micro_step``."""
    names = parameter_order(cfg)
    spec = state_dict_spec(cfg)
    osd = checkpoint["optimizer_states"][0]
    n_ckpt = sum(len(g["params"]) for g in osd["param_groups"])
    if n_ckpt != len(names):
        raise ValueError(f"the checkpoint's optimizer holds {n_ckpt} parameters, this architecture has {len(names)} "
                         "(a FastDiff vocoder attached to the model adds its own)")
    st = osd["state"]
    steps = set()
    out = {"exp_avg": {}, "exp_avg_sq": {}}
    for i, n in enumerate(names):
        if n.endswith(".bins"):
            continue  # frozen: a slot in the numbering, no state
        ent = st.get(i, st.get(str(i)))
        if ent is None:  # no step taken for this parameter yet
            out["exp_avg"][n] = torch.zeros(spec[n])
            out["exp_avg_sq"][n] = torch.zeros(spec[n])
            continue
        for key in ("exp_avg", "exp_avg_sq"):
            t = torch.as_tensor(ent[key]).detach().cpu().to(torch.float32)
            if tuple(t.shape) != tuple(spec[n]):
                raise ValueError(f"optimizer state {key}[{i}] has shape {tuple(t.shape)}; parameter {i} of this architecture is "
                                 f"{n} {tuple(spec[n])}")
            out[key][n] = t
        steps.add(int(float(ent["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state this optimizer can resume")
    sched = (checkpoint.get("lr_schedulers") or [None])[0]
    out["step"] = steps.pop() if steps else int((sched or {}).get("last_epoch", checkpoint.get("global_step", 0)))
    # the dropout seed of a micro-batch is seed * 1000003 + micro_step (training.Trainer): resuming at 0 would replay the
    # masks of steps 0..N.  Lightning keeps no such counter; optimizer steps x accumulate_grad_batches is what it would be.
    out["micro_step"] = int(checkpoint.get("fs2_micro_step", out["step"] * max(1, int(accumulate_grad_batches))))
    return out
