"""Host-side mirror of the reference's model interface for the mel forward.

``FastSpeech2`` here plays the role of ``litfass.fastspeech2.fastspeech2.FastSpeech2`` for its
callers on the inference path — ``SpeechGenerator.generate_samples`` does
``self.model(batch, inference=True)`` and reads ``mel`` / ``tgt_mask`` (synthesis/generator.py:158-165)
and ``generate.py`` reads ``.hparams``, ``.phone2id``, ``.speaker2dvector``, ``.stats``, ``.device``
(generate.py:139-151,200-204) — with the same dict-in / dict-out contract as
``FastSpeech2.forward`` (fastspeech2.py:636-784).  All arithmetic happens in libfs2_hip.so (hand-written
gfx950 kernels) through the C ABI in include/fs2.h; PyTorch is used for device memory and streams
only.  There is no CPU fallback: without the HIP library or a GPU this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import Fs2Config

# "mixed": fp32 for everything a discrete decision hangs on (encoder, durations, variance predictors / buckets),
# bf16 for the decoder + mel head (include/fs2.h: FS2_MIXED)
# "mixed3": the same with the front's matrix products as bf16 x 3 split products of the fp32 operands (FS2_MIXED_X3)
_PRECISIONS = {"fp32": _lib.FS2_F32, "f32": _lib.FS2_F32, "bf16": _lib.FS2_BF16, "mixed": _lib.FS2_MIXED,
               "mixed3": _lib.FS2_MIXED_X3, "fp32x3": _lib.FS2_F32_X3}
# "fp32x3": the fp32 mode's storage, attention, LayerNorm, heads and decision logic with EVERY GEMM / conv as bf16 x 3 split products
# (FS2_F32_X3): ~1e-5 on the mel against "fp32" at about half its time


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """Thin owner of one ``fs2_engine`` handle (one per device)."""

    def __init__(self, cfg: Fs2Config, state_dict, precision: str = "fp32", device="cuda:0",
                 torch_workspace: bool = True, graphs: Optional[bool] = None):
        self.lib = _lib.load()
        # torch_workspace: the forward's device workspace comes from torch's caching allocator through
        # fs2_workspace_bytes / fs2_set_workspace (caller-owned memory); False = the
        # engine's own grow-only hipMalloc arenas
        self.torch_workspace = torch_workspace
        self._ws_persist = self._ws_scratch = None
        if not torch.cuda.is_available():
            raise RuntimeError("lightningfastspeech2_amd needs an MI355X (no CPU fallback for the product path)")
        self.cfg = cfg
        self.precision = precision
        self.dtype = _PRECISIONS[precision]
        self.device = torch.device(device)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            cc = _lib.config_to_c(cfg, self.dtype)
            st = self.lib.fs2_create(C.byref(cc), C.byref(self.handle))
            try:
                _lib.check(st, self.handle, "create")
                self._load(state_dict)
            except Exception:
                self.close()
                raise
        self._last = None
        self._t_hint = {}
        # Both phases of the forward can replay as hipGraphs once a shape + buffer signature repeats (first sight plain, second
        # captured): the step then costs the host two graph launches around its one sync instead of ~150 kernel launches, so a
        # busy host no longer shows up as GPU idle time.  Measured on a quiet host (r03, C2 batch 32): eager 2.37 ms / step =
        # the forward's GPU time to 0.4 %, replay 2.48 (ROCm's graph launch costs ~0.7 us per node) - so eager is the default
        # and graphs are the caller's choice (graphs=True / FS2_GRAPHS=1); bench.py times a few steps of both during warm-up
        # and runs the timed region in the faster mode of the box it is on.
        if graphs is None:
            graphs = os.environ.get("FS2_GRAPHS", "0") == "1"
        self.set_graphs(graphs)

    def clone(self) -> "Engine":
        """Another engine over the same device weights (fs2_clone): own workspace and host-side state."""
        other = Engine.__new__(Engine)
        other.lib, other.torch_workspace = self.lib, self.torch_workspace
        other._ws_persist = other._ws_scratch = None
        other.cfg, other.precision, other.dtype, other.device = self.cfg, self.precision, self.dtype, self.device
        other.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fs2_clone(self.handle, C.byref(other.handle)), self.handle, "clone")
        other._last, other._t_hint = None, {}
        other._graphs_on = getattr(self, "_graphs_on", False)
        return other

    def _load(self, state_dict):
        from .weights import state_dict_spec
        spec = state_dict_spec(self.cfg)
        for name in spec:
            if name not in state_dict:
                raise KeyError(f"state_dict is missing '{name}'")
            v = state_dict[name]
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().float().numpy()
            a = np.ascontiguousarray(v, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            st = self.lib.fs2_load_weight(self.handle, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
            _lib.check(st, self.handle, f"load_weight({name})")
        _lib.check(self.lib.fs2_finalize(self.handle), self.handle, "finalize")

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.fs2_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_debug(self, on: bool):
        _lib.check(self.lib.fs2_set_debug(self.handle, int(on)), self.handle, "set_debug")

    def encode(self, phones: torch.Tensor, speaker: torch.Tensor, forced_durations: Optional[torch.Tensor] = None,
               priors: Optional[torch.Tensor] = None) -> int:
        B, L = phones.shape
        T = C.c_int32(0)
        if priors is not None:  # (n_priors, B) fp32 on the device
            _lib.check(self.lib.fs2_set_priors(self.handle, _ptr(priors), B), self.handle, "set_priors")
        with torch.cuda.device(self.device):
            if self.torch_workspace:  # sized for the frame count this (B, L) produced last time, if any
                self._ensure_workspace(B, L, self._t_hint.get((B, L), 0))
            st = self.lib.fs2_encode(self.handle, _ptr(phones), _ptr(speaker), B, L, _ptr(forced_durations),
                                     self._stream(), C.byref(T))
        _lib.check(st, self.handle, "encode")
        self._last = (B, L, T.value)
        self._t_hint[(B, L)] = T.value
        return T.value

    def workspace_bytes(self, B: int, L: int, T: int = 0):
        """(persist_bytes, scratch_bytes) a (B, L) encode and a T-frame decode need (fs2_workspace_bytes)."""
        pb, sb = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.fs2_workspace_bytes(self.handle, B, L, T, C.byref(pb), C.byref(sb)), self.handle,
                   "workspace_bytes")
        return pb.value, sb.value

    def _ensure_workspace(self, B: int, L: int, T: int):
        pb, sb = self.workspace_bytes(B, L, T)
        grow_p = self._ws_persist is None or self._ws_persist.numel() < pb
        grow_s = self._ws_scratch is None or self._ws_scratch.numel() < sb
        if not (grow_p or grow_s):
            return
        # torch's allocator returns >= 512-byte aligned blocks and keeps freed blocks alive until the
        # work queued on their stream has run, so replacing a buffer is safe without a device sync
        if grow_p:
            self._ws_persist = torch.empty(pb + pb // 8, dtype=torch.uint8, device=self.device)
        if grow_s:
            self._ws_scratch = torch.empty(sb + sb // 8, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.fs2_set_workspace(self.handle, _ptr(self._ws_persist), self._ws_persist.numel(),
                                              _ptr(self._ws_scratch), self._ws_scratch.numel()), self.handle,
                   "set_workspace")

    def set_frames(self, T: int):
        """Pad this shard's decode to T frames (>= its own): the data-parallel global-pad mode."""
        _lib.check(self.lib.fs2_set_frames(self.handle, int(T)), self.handle, "set_frames")
        B, L, _ = self._last
        self._last = (B, L, int(T))

    def set_zero_pad_mel(self, on: bool):
        _lib.check(self.lib.fs2_set_zero_pad_mel(self.handle, int(on)), self.handle, "set_zero_pad_mel")

    def totals(self):
        B = self._last[0]
        tot = np.zeros(B, np.int32)
        grd = np.zeros(B, np.int32)
        _lib.check(self.lib.fs2_last_totals(self.handle, tot.ctypes.data_as(C.c_void_p),
                                            grd.ctypes.data_as(C.c_void_p), B), self.handle, "last_totals")
        return tot, grd

    def force_buckets(self, var_index: int, idx: torch.Tensor):
        """Next decode embeds these (B, T) int32 bucket indices for that variance (parity aid)."""
        self._keep = getattr(self, "_keep", []) + [idx]
        _lib.check(self.lib.fs2_force_buckets(self.handle, var_index, _ptr(idx)), self.handle, "force_buckets")

    def alloc_outputs(self, B: int, L: int, T: int, want_aux: bool = True) -> Dict[str, torch.Tensor]:
        """Fresh output tensors for a (B, L, T) forward (the reference returns fresh tensors too)."""
        dev = self.device
        res = {"mel": torch.empty(B, T, self.cfg.n_mels, dtype=torch.float32, device=dev)}
        if want_aux:
            res["duration_prediction"] = torch.empty(B, L, dtype=torch.float32, device=dev)
            res["duration_rounded"] = torch.empty(B, L, dtype=torch.int32, device=dev)
            res["src_mask"] = torch.empty(B, L, dtype=torch.bool, device=dev)
            res["tgt_mask"] = torch.empty(B, T, dtype=torch.bool, device=dev)
            for i, var in enumerate(self.cfg.variances):
                S = L if self.cfg.is_phone_level(i) else T  # a phone-level variance is predicted per phone (model.py:276-294)
                res[f"variances_{var}"] = torch.empty(B, S, dtype=torch.float32, device=dev)
                if self.cfg.is_cwt(i):  # by-products of the CWT head (model.py:445-461)
                    res[f"_cwt_spectrogram_{var}"] = torch.empty(B, S, 10, dtype=torch.float32, device=dev)
                    res[f"_cwt_mean_std_{var}"] = torch.empty(B, 2, dtype=torch.float32, device=dev)
        return res

    def force_variance_targets(self, var_index: int, tgt: torch.Tensor):
        """Next decode embeds bucketize(tgt*std+mean) for that variance (teacher forcing)."""
        self._keep = getattr(self, "_keep", []) + [tgt]
        _lib.check(self.lib.fs2_force_variance_targets(self.handle, var_index, _ptr(tgt)), self.handle,
                   "force_variance_targets")

    def decode(self, want_aux: bool = True, outputs: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        B, L, T = self._last
        if self.torch_workspace:
            with torch.cuda.device(self.device):
                self._ensure_workspace(B, L, T)  # only the scratch buffer can grow here
        res = outputs
        if res is None or tuple(res["mel"].shape) != (B, T, self.cfg.n_mels):
            res = self.alloc_outputs(B, L, T, want_aux)
        out = _lib.Fs2OutputsC()
        out.mel = res["mel"].data_ptr()
        if "tgt_mask" in res:
            out.duration_prediction = res["duration_prediction"].data_ptr()
            out.duration_rounded = res["duration_rounded"].data_ptr()
            out.src_mask = res["src_mask"].data_ptr()
            out.tgt_mask = res["tgt_mask"].data_ptr()
            for i, var in enumerate(self.cfg.variances):
                out.variances[i] = res[f"variances_{var}"].data_ptr()
                if self.cfg.is_cwt(i):
                    out.var_spectrogram[i] = res[f"_cwt_spectrogram_{var}"].data_ptr()
                    out.var_mean_std[i] = res[f"_cwt_mean_std_{var}"].data_ptr()
        with torch.cuda.device(self.device):
            st = self.lib.fs2_decode(self.handle, C.byref(out), self._stream())
        _lib.check(st, self.handle, "decode")
        self._keep = []
        return res

    def debug_tensor(self, what: str) -> torch.Tensor:
        B, L, T = self._last
        H = self.cfg.hidden
        if what == "encoder_out":
            t = torch.empty(B, L, H, dtype=torch.float32, device=self.device)
        elif what.startswith("bucket_"):
            var = what[len("bucket_"):]
            phone = var in self.cfg.variances and self.cfg.is_phone_level(self.cfg.variances.index(var))
            t = torch.empty(B, L if phone else T, dtype=torch.int32, device=self.device)
        else:
            t = torch.empty(B, T, H, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fs2_debug_copy(self.handle, what.encode(), _ptr(t), self._stream()), self.handle,
                   f"debug_copy({what})")
        return t

    def set_deferred_layernorm(self, on: bool):
        """A/B: wide depth-wise blocks with LayerNorm deferred into its consumers (default) or as its own launches."""
        _lib.check(self.lib.fs2_set_deferred_layernorm(self.handle, int(on)), self.handle, "set_deferred_layernorm")

    def set_tuning(self, knob: int):
        """One kernel-selection switch of this engine (include/fs2.h, the values of fs2_op_set_gemm_variant): A/B runs and the tests
        that pin two forms of a kernel against each other.  Replicas made afterwards (``FastSpeech2.replicate``) inherit it."""
        _lib.check(self.lib.fs2_set_tuning(self.handle, int(knob)), self.handle, f"set_tuning({knob})")

    def set_folded_layernorm(self, on: bool):
        """A/B: inside a stack of wide depth-wise bf16 blocks, a block's closing LayerNorm folded into the next block's in-projection
        (default) or one normalise-only pass per block."""
        _lib.check(self.lib.fs2_set_folded_layernorm(self.handle, int(on)), self.handle, "set_folded_layernorm")

    def set_graphs(self, on: bool):
        """Replay the encode and the decode phase (~100 / ~50 launches) as hipGraphs once a shape + buffer signature repeats
        (include/fs2.h fs2_set_graphs); bit-identical results, the forward stops depending on how fast the host can issue
        launches.  Off by default (constructor argument graphs=True / FS2_GRAPHS=1)."""
        _lib.check(self.lib.fs2_set_graphs(self.handle, int(on)), self.handle, "set_graphs")
        self._graphs_on = bool(on)

    def graph_replays(self) -> int:
        return int(self.lib.fs2_graph_replays(self.handle))

    def set_fused_predictor(self, on: bool):
        """A/B: run the variance/duration predictors as one launch each (default) or layer by layer."""
        _lib.check(self.lib.fs2_set_fused_predictor(self.handle, int(on)), self.handle, "set_fused_predictor")

    def profile_enable(self, kernel_class: int, on: bool = True):
        _lib.check(self.lib.fs2_profile_enable(self.handle, kernel_class, int(on)), self.handle, "profile_enable")

    def profile_reserve(self, kernel_class: int, pairs: int):
        """create the event pairs of `pairs` bracketed launches ahead of a timed region"""
        _lib.check(self.lib.fs2_profile_reserve(self.handle, kernel_class, int(pairs)), self.handle, "profile_reserve")

    def profile_read(self, kernel_class: int):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self.lib.fs2_profile_read(self.handle, kernel_class, C.byref(ms), C.byref(n), C.byref(fl),
                                             C.byref(by)), self.handle, "profile_read")
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}


class FastSpeech2:
    """Drop-in for the reference model object on the inference path.

    ``model(batch, inference=True)`` returns the same dict as the reference: ``mel`` (B,T,n_mels)
    fp32 including pad rows, ``duration_prediction``, ``duration_rounded`` (int32), ``src_mask`` /
    ``tgt_mask`` (bool, True = pad) and ``variances_<var>``; tensors live on ``self.device``.
    """

    def __init__(self, cfg: Fs2Config, state_dict, *, precision: str = "fp32", device="cuda:0",
                 phone2id: Optional[dict] = None, speaker2dvector: Optional[dict] = None,
                 extra_hparams: Optional[dict] = None):
        self.cfg = cfg
        self.stats = cfg.stats
        self.phone2id = phone2id
        self.speaker2dvector = speaker2dvector
        hp = cfg.to_dict()
        hp.pop("stats", None)
        hp.pop("n_phones", None)
        hp.update(dict(speaker_embedding_every_layer=False, prior_embedding_every_layer=False,
                       fastdiff_variances=False, fastdiff_speakers=False, duration_stochastic=False))
        hp.update(extra_hparams or {})
        self.hparams = SimpleNamespace(**hp)
        self.engine = Engine(cfg, state_dict, precision=precision, device=device)
        self.device = self.engine.device
        self.training = False
        self._t_guess = {}

    def replicate(self) -> "FastSpeech2":
        """Another model object over the same weights (its own engine: own workspace, own host state) - what ForwardPipeline
        keeps per forward in flight."""
        import copy
        other = copy.copy(self)            # hparams, stats, phone2id ...: shared, read-only
        other.engine = self.engine.clone()  # the device weights too (fs2_clone)
        other._t_guess = {}
        return other

    def pipeline(self, in_flight: int = 2, host_outputs=()) -> "ForwardPipeline":
        return ForwardPipeline(self, in_flight, host_outputs)

    # ---- reference-style construction from a Lightning checkpoint dict (fastspeech2.py:530-634) --
    @classmethod
    def from_checkpoint(cls, checkpoint, *, precision: str = "fp32", device="cuda:0", tolerant: bool = False,
                        init_seed: int = 0):
        """``FastSpeech2.load_from_checkpoint(path)`` (generate.py:112) for a ``lit_model.ckpt`` path or an already
        loaded checkpoint dict.  ``tolerant=True`` reproduces the reference's shape-mismatch-tolerant load
        (fastspeech2.py:598-620; see checkpoint.resolve_state_dict), the default refuses a tensor that does not fit."""
        from . import checkpoint as ck
        if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
            checkpoint = ck.read_checkpoint(checkpoint)
        cfg = ck.config_from_checkpoint(checkpoint)
        sd, report = ck.resolve_state_dict(cfg, checkpoint["state_dict"], tolerant=tolerant, init_seed=init_seed)
        model = cls(cfg, sd, precision=precision, device=device, phone2id=checkpoint["phone2id"],
                    speaker2dvector=checkpoint.get("speaker2dvector"))
        model.load_report = report
        for extra in ck.EXTRA_KEYS:  # fastspeech2.py:571-587
            if extra in checkpoint and extra != "speaker2dvector":
                setattr(model, extra, checkpoint[extra])
        return model

    # ---- nn.Module-ish surface the callers touch (generator.py:58-62) -----------------------
    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("an fs2 engine is bound to the device it was created on; build a new "
                               "FastSpeech2(..., device=...) for another GPU")
        return self

    def _target_key(self, vi: int) -> str:
        """Teacher-forcing target of variance vi: the raw signal for a CWT variance (model.py:319-321)."""
        var = self.cfg.variances[vi]
        return f"variances_{var}_signal" if self.cfg.is_cwt(vi) else f"variances_{var}"

    def __call__(self, targets, inference: bool = False):
        return self.forward(targets, inference)

    def forward(self, targets: dict, inference: bool = False, *, force_durations=None, force_buckets=None,
                frames_hook=None) -> dict:
        """``frames_hook(T_local) -> T`` (optional) runs between the two phases and may raise the frame count
        this batch is padded to (data-parallel global-pad mode: an all-reduce MAX over the ranks)."""
        teacher = not inference and force_durations is None
        if teacher:
            # FastSpeech2.forward(batch) as the Lightning hooks call it (fastspeech2.py:787,800):
            # target durations (model.py:296-297) and target variances (model.py:317-325) are teacher
            # forced; forward only — the loss/backward of training_step stay with the caller.
            missing = [k for k in ["duration"] + [self._target_key(i) for i in range(len(self.cfg.variances))] if k not in targets]
            if missing:
                raise KeyError(f"inference=False needs teacher-forcing targets {missing} (model.py:296-297,317-325)")
            force_durations = targets["duration"]
        phones = targets["phones"]
        speaker = targets["speaker"]
        if not isinstance(phones, torch.Tensor):
            phones = torch.as_tensor(np.asarray(phones))
        if not isinstance(speaker, torch.Tensor):
            speaker = torch.as_tensor(np.asarray(speaker))
        if phones.dim() != 2 or speaker.dim() != 2 or speaker.shape[0] != phones.shape[0]:
            raise ValueError("phones must be (B, L) and speaker (B, 256)")
        if not phones.is_cuda:  # same failure the reference's nn.Embedding raises on a bad id
            if phones.numel() and (int(phones.min()) < 0 or int(phones.max()) >= self.cfg.n_phones):
                raise IndexError("phone id out of range")
        phones = phones.to(self.device, dtype=torch.int64).contiguous()      # fastspeech2.py:639
        speaker = speaker.to(self.device, dtype=torch.float32).contiguous()  # fastspeech2.py:641
        forced = None
        if force_durations is not None:
            forced = torch.as_tensor(force_durations)
            if forced.dim() != 2 or forced.shape[0] != phones.shape[0] or forced.shape[1] < phones.shape[1]:
                raise ValueError(f"durations must be (B, L) = {tuple(phones.shape)}, got {tuple(forced.shape)}")
            if forced.shape[1] > phones.shape[1]:  # a wider target is fine only if the surplus is padding
                if bool((forced[:, phones.shape[1]:] != 0).any()):
                    raise ValueError(f"durations {tuple(forced.shape)} carry non-zero entries beyond phones' L={phones.shape[1]}")
                forced = forced[:, :phones.shape[1]]
            forced = forced.to(self.device, dtype=torch.int32).contiguous()
        # Output buffers are allocated BEFORE the forward's one host sync (inside fs2_encode), sized
        # with the frame count of the previous call: in steady state (same T) the allocator work
        # overlaps the encoder instead of sitting between the two phases; a different T just
        # allocates again.  Fresh tensors every call, like the reference.
        B, L = phones.shape
        guess = self._t_guess.get((B, L))
        pre = self.engine.alloc_outputs(B, L, guess) if guess else None
        priors = None
        if self.cfg.priors:  # utterance-level priors, fastspeech2.py:687-692
            # generate_from_text hands these over as device tensors (generator.py:131-146), datasets as lists/arrays
            rows = [(v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float32)))
                    .to(self.device, dtype=torch.float32).reshape(B) for v in (targets[f"priors_{pr}"] for pr in self.cfg.priors)]
            priors = torch.stack(rows).contiguous()
        # phone-level variances are embedded by the encode phase (model.py:276-294): their teacher-forcing targets / forced buckets
        # are (B, L) and go in BEFORE it
        for vi, var in enumerate(self.cfg.variances):
            if not self.cfg.is_phone_level(vi):
                continue
            if teacher:
                key = self._target_key(vi)
                tgt = torch.as_tensor(np.asarray(targets[key]) if not isinstance(targets[key], torch.Tensor)
                                      else targets[key]).to(self.device, dtype=torch.float32)
                if tgt.dim() != 2 or tgt.shape[0] != B or tgt.shape[1] < L:
                    raise ValueError(f"targets[{key!r}] (phone level) must be (B, >= L) = ({B}, >= {L})")
                if self.cfg.is_cwt(vi):
                    tgt = torch.log(tgt)
                self.engine.force_variance_targets(vi, tgt[:, :L].contiguous())
            if force_buckets and var in force_buckets:
                idx = torch.as_tensor(force_buckets[var]).to(self.device, dtype=torch.int32).contiguous()
                if tuple(idx.shape) != (B, L):
                    raise ValueError(f"force_buckets[{var!r}] (phone level) must be (B, L)=({B}, {L})")
                self.engine.force_buckets(vi, idx)
        T = self.engine.encode(phones, speaker, forced, priors)
        if frames_hook is not None:
            Tg = int(frames_hook(T))
            if Tg != T:
                self.engine.set_frames(Tg)
                T = Tg
        self._t_guess[(B, L)] = T
        if teacher:
            for vi, var in enumerate(self.cfg.variances):
                if self.cfg.is_phone_level(vi):
                    continue
                key = self._target_key(vi)
                tgt = torch.as_tensor(np.asarray(targets[key]) if not isinstance(targets[key], torch.Tensor)
                                      else targets[key]).to(self.device, dtype=torch.float32)
                if tgt.dim() != 2 or tgt.shape[0] != B or tgt.shape[1] < T:
                    raise ValueError(f"targets[{key!r}] must be (B, >= T) = ({B}, >= {T})")
                if self.cfg.is_cwt(vi):  # tgt = torch.log(tgt), bucketised as it is (model.py:418-419)
                    tgt = torch.log(tgt)
                self.engine.force_variance_targets(vi, tgt[:, :T].contiguous())
        for var, idx in (force_buckets or {}).items():
            if self.cfg.is_phone_level(self.cfg.variances.index(var)):
                continue  # handed over before the encode phase
            idx = torch.as_tensor(idx).to(self.device, dtype=torch.int32).contiguous()
            if tuple(idx.shape) != (phones.shape[0], T):
                raise ValueError(f"force_buckets[{var!r}] must be (B, T)=({phones.shape[0]}, {T})")
            self.engine.force_buckets(self.cfg.variances.index(var), idx)
        res = self.engine.decode(outputs=pre)
        for vi, var in enumerate(self.cfg.variances):  # the CWT head hands back a dict (model.py:445-461)
            if self.cfg.is_cwt(vi):
                ms = res.pop(f"_cwt_mean_std_{var}")
                d = {"spectrogram": res.pop(f"_cwt_spectrogram_{var}"), "mean": ms[:, 0], "std": ms[:, 1]}
                sig = res[f"variances_{var}"]
                res[f"variances_{var}"] = d if teacher else {"reconstructed_signal": torch.exp(sig), **d}
        if teacher:  # the reference hands back the target tensor itself (model.py:297,337)
            d = targets["duration"]
            res["duration_rounded"] = d.to(self.device) if isinstance(d, torch.Tensor) else torch.as_tensor(np.asarray(d)).to(self.device)
        _, guard = self.engine.totals()
        for _ in range(int(guard.sum())):
            print("Zero duration, setting to 1")  # the reference's one stdout side effect (model.py:309)
        return res


class ForwardPipeline:
    """``in_flight`` inference forwards at once: batch i + 1's encoder is already queued while the host reads batch i's frame
    count - the forward's one host sync (fastspeech2.py / model.py:354-355: the output length is data dependent) - and launches
    its decoder.

    A single forward leaves the GPU idle at that seam (sync wake-up, output allocation, the first decode launches) and in the
    tail / ramp of every launch that does not fill the chip; a second forward on its own HIP stream fills both.  One engine
    replica (own workspace arenas, own host-side state), one HIP stream and one host thread per forward in flight; ctypes
    releases the GIL inside the library, so the threads drive their streams concurrently.  Outputs are bit-identical to
    ``model(batch, inference=True)`` (same engine code, same kernels; tests/test_gpu_boundary.py).  Measured r04 at C2, batch 32:
    2.06 -> ~1.9 ms per batch with two in flight (profiles/HISTORY.md §4).

        pipe = model.pipeline(2)
        for batch in batches:
            for out in pipe.submit(batch):   # results come back in submission order, a batch or two later
                use(out)
        for out in pipe.drain():
            use(out)

    A result's tensors were produced on the pipeline's own streams; ``submit`` / ``drain`` make the caller's current stream wait for
    them (an event per forward) before handing them over.

    ``host_outputs`` (r06; e.g. ``("mel", "tgt_mask")``): the host boundary inside the pipeline - what
    ``SpeechGenerator.generate_samples`` does right behind the forward (generator.py:158-165: ``mel[i][~tgt_mask[i]].cpu()``).  A batch
    may then be HOST tensors (pinned: copied to the device on the replica's upload stream, which the forward's stream waits for), and the named outputs come back
    as PINNED HOST tensors: their device-to-host copies are queued on a per-replica copy stream behind the forward, so batch i's 15.7 MB
    of mels cross PCIe while batch i + 1's forward runs.  The bytes are those of ``model(batch, inference=True)[key].cpu()``.  A host
    output is a view of a per-replica ring slot (four per replica, up to 2 x in_flight results pending): it stays valid for the next
    ``in_flight`` calls of ``submit`` after it was handed over (copy it or consume it before); the other outputs stay device tensors.
    """

    def __init__(self, model: FastSpeech2, in_flight: int = 2, host_outputs=()):
        import concurrent.futures as cf
        if in_flight < 1:
            raise ValueError("in_flight >= 1")
        self.host_outputs = tuple(host_outputs)
        self._ring = [[{} for _ in range(4)] for _ in range(in_flight)]  # [replica][slot] -> {key: flat pinned uint8 buffer}
        self._nrun = [0] * in_flight
        self.models = [model] + [model.replicate() for _ in range(in_flight - 1)]
        for m in self.models[1:]:
            m.engine.set_graphs(getattr(model.engine, "_graphs_on", False))
        self.device = model.device
        self.streams = [torch.cuda.Stream(self.device) for _ in self.models]
        self.copy_streams = [torch.cuda.Stream(self.device) for _ in self.models] if self.host_outputs else []
        self.up_streams = [torch.cuda.Stream(self.device) for _ in self.models] if self.host_outputs else []
        self.copy_pools = [cf.ThreadPoolExecutor(max_workers=1) for _ in self.models] if self.host_outputs else []
        # one single-thread executor per replica: a replica's forwards run in submission order on its own thread and stream
        self.pools = [cf.ThreadPoolExecutor(max_workers=1) for _ in self.models]
        self.pending = []  # futures in submission order
        self.n = 0

    def set_graphs(self, on: bool):
        for m in self.models:
            m.engine.set_graphs(on)

    @staticmethod
    def _record(obj, stream):
        """Tell torch's caching allocator that `stream` uses these tensors too: a block is otherwise returned to the pool of the
        stream it was allocated on as soon as its tensor dies, and the next allocation there may overwrite it while the other
        stream still reads it (a mel dropped by the caller right after queueing a vocoder on it; a batch released by
        the caller while stream k still decodes from it)."""
        if isinstance(obj, torch.Tensor):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, dict):
            for v in obj.values():
                ForwardPipeline._record(v, stream)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                ForwardPipeline._record(v, stream)

    def _run(self, k, batch, ready):
        with torch.cuda.device(self.device), torch.cuda.stream(self.streams[k]):
            self.streams[k].wait_event(ready)  # the caller's stream produced the batch
            self._record(batch, self.streams[k])  # ... and may release it before this stream has finished reading it
            if self.host_outputs:
                # host inputs go up on the replica's own (otherwise idle) upload stream: the runtime performs a SMALL copy synchronously with
                # the host once the stream reaches it (measured r06, tools/probes/pcie_pipeline_probe.py: 66 KB of inputs queued behind the
                # replica's previous forward cost +0.26 ms per batch) - on an idle stream that wait is the copy itself
                with torch.cuda.stream(self.up_streams[k]):
                    moved = {key: v.to(self.device, non_blocking=True) for key, v in batch.items()
                             if isinstance(v, torch.Tensor) and not v.is_cuda and key in ("phones", "speaker")}
                    up = torch.cuda.Event()
                    up.record(self.up_streams[k])
                self.streams[k].wait_event(up)
                self._record(moved, self.streams[k])
                batch = {**batch, **moved}
            out = self.models[k](batch, inference=True)
            done = torch.cuda.Event()
            done.record(self.streams[k])
            if self.host_outputs:
                # the device-to-host copies are queued by the replica's COPIER thread: the small ones (a 49 KB mask) block their caller until
                # the copy stream reaches them, i.e. until this forward has finished - the forward thread must be back for the next batch by then
                return self.copy_pools[k].submit(self._to_host, k, out, done), None
        return out, done

    def _to_host(self, k, out, fwd_done):
        """Queue the device-to-host copies of the named outputs on replica k's copy stream (behind the forward, beside the next one)."""
        cs = self.copy_streams[k]
        slot = self._ring[k][self._nrun[k] % 4]
        self._nrun[k] += 1
        cs.wait_event(fwd_done)
        res = dict(out)
        with torch.cuda.device(self.device), torch.cuda.stream(cs):
            for key in self.host_outputs:
                src = out[key]
                if not isinstance(src, torch.Tensor) or not src.is_cuda:
                    continue
                nbytes = src.numel() * src.element_size()
                buf = slot.get(key)
                if buf is None or buf.numel() < nbytes:
                    buf = slot[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8).pin_memory()
                dst = buf[:nbytes].view(src.dtype).view(src.shape)
                dst.copy_(src.contiguous(), non_blocking=True)
                src.record_stream(cs)
                res[key] = dst
            done = torch.cuda.Event()
            done.record(cs)
        return res, done

    def _hand_over(self, fut):
        out, done = fut.result()
        if done is None:  # host outputs: the copier's future
            out, done = out.result()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        self._record(out, cur)  # allocated on the pipeline's stream, consumed (and possibly dropped) on the caller's
        if self.host_outputs:
            done.synchronize()  # a host tensor is read by the host: the copy has landed (what .cpu() waits for in the reference's caller)
        return out

    def submit(self, batch):
        """Queue one batch; returns the list (possibly empty) of finished-in-order results that fall out of the window."""
        k = self.n % len(self.models)
        self.n += 1
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self.pending.append(self.pools[k].submit(self._run, k, batch, ready))
        outs = []
        # never more than in_flight behind: the host waits for the oldest.  With host outputs a result is finished only when its copy has
        # landed, which runs BESIDE the forwards (its replica is already on its next batch): the window counts one more round, so that
        # waiting for batch i's copy does not keep batch i + in_flight + 1 from being queued
        window = len(self.models) * (2 if self.host_outputs else 1)
        while len(self.pending) > window:
            outs.append(self._hand_over(self.pending.pop(0)))
        while self.pending and self.pending[0].done():     # and whatever has finished in the meantime
            if self.host_outputs:  # finished = its host copies have landed
                cf_ = self.pending[0].result()[0]
                if not cf_.done() or not cf_.result()[1].query():
                    break
            outs.append(self._hand_over(self.pending.pop(0)))
        return outs

    def drain(self):
        outs = [self._hand_over(f) for f in self.pending]
        self.pending = []
        return outs

    def close(self):
        self.drain()
        for p in self.pools + self.copy_pools:
            p.shutdown(wait=True)
