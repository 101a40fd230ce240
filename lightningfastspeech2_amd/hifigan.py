"""Host mirror of the reference's HiFi-GAN vocoder wrapper.

Reference interface: ``litfass.third_party.hifigan.Synthesiser`` (third_party/hifigan/__init__.py:19-43):
``Synthesiser(device, model)`` loads ``generator_<model>.pth.tar``, removes weight norm, and
``synth(mel)`` maps ONE utterance's mel ``(T, 80)`` to int16 samples ``(1, T*256)``
(called per utterance by ``SpeechGenerator.generate_samples``, synthesis/generator.py:163-170).
The arithmetic (``Generator.forward``, third_party/hifigan/models.py:145-162) runs in
``libfs2_hip.so`` through the ``fs2_voc_*`` C ABI; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib


@dataclass
class HifiGanConfig:
    """The fields of third_party/hifigan/config.json that shape Generator.forward."""
    resblock: str = "1"
    upsample_rates: List[int] = field(default_factory=lambda: [8, 8, 2, 2])
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 16, 4, 4])
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[List[int]] = field(default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    num_mels: int = 80
    sampling_rate: int = 22050

    def __post_init__(self):
        if str(self.resblock) != "1":
            raise ValueError("only resblock type '1' (the shipped config) is implemented")
        if len(self.upsample_rates) != len(self.upsample_kernel_sizes):
            raise ValueError("upsample_rates / upsample_kernel_sizes length mismatch")
        if len(self.resblock_kernel_sizes) != len(self.resblock_dilation_sizes):
            raise ValueError("resblock_kernel_sizes / resblock_dilation_sizes length mismatch")
        for u, k in zip(self.upsample_rates, self.upsample_kernel_sizes):
            if (k - u) % 2 or k - 2 * ((k - u) // 2) != u:
                raise ValueError("ConvTranspose1d kernel - 2*padding must equal the stride (models.py:131-136)")
        if (self.upsample_initial_channel >> len(self.upsample_rates)) % 32:
            raise ValueError("the last stage must keep a multiple of 32 channels")

    @property
    def hop(self) -> int:
        return int(np.prod(self.upsample_rates))

    def channels(self) -> List[int]:
        return [self.upsample_initial_channel >> i for i in range(len(self.upsample_rates) + 1)]

    @classmethod
    def from_json(cls, text_or_dict) -> "HifiGanConfig":
        d = json.loads(text_or_dict) if isinstance(text_or_dict, str) else dict(text_or_dict)
        return cls(**{k: d[k] for k in cls.__dataclass_fields__ if k in d})


def state_dict_spec(cfg: HifiGanConfig) -> Dict[str, tuple]:
    """Generator.state_dict() names/shapes after remove_weight_norm (models.py:116-143)."""
    ch = cfg.channels()
    s = {"conv_pre.weight": (ch[0], cfg.num_mels, 7), "conv_pre.bias": (ch[0],)}
    nk = len(cfg.resblock_kernel_sizes)
    for i, k in enumerate(cfg.upsample_kernel_sizes):
        s[f"ups.{i}.weight"] = (ch[i], ch[i + 1], k)  # ConvTranspose1d: (in, out, k)
        s[f"ups.{i}.bias"] = (ch[i + 1],)
        for j, rk in enumerate(cfg.resblock_kernel_sizes):
            for grp in ("convs1", "convs2"):
                for m in range(3):
                    s[f"resblocks.{i * nk + j}.{grp}.{m}.weight"] = (ch[i + 1], ch[i + 1], rk)
                    s[f"resblocks.{i * nk + j}.{grp}.{m}.bias"] = (ch[i + 1],)
    s["conv_post.weight"] = (1, ch[-1], 7)
    s["conv_post.bias"] = (1,)
    return s


def synth_state_dict(cfg: HifiGanConfig, seed: int = 0, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Random generator weights (there is no network for generator_universal.pth.tar): N(0, gain/fan_in)
    conv weights so that activations stay O(1) through the 4 x 3 x 6 conv stack, small biases.
    numpy RandomState -> identical on every host."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, shape in state_dict_spec(cfg).items():
        if name.endswith(".bias"):
            sd[name] = (0.05 * rs.standard_normal(shape)).astype(np.float32)
        elif name.startswith("ups."):
            cin, _, k = shape
            u = cfg.upsample_rates[int(name.split(".")[1])]
            sd[name] = (rs.standard_normal(shape) * gain * (u / (cin * k)) ** 0.5).astype(np.float32)
        else:
            _, cin, k = shape
            g = gain * (0.1 if name.startswith("conv_post") else 1.0)  # keep tanh out of saturation
            sd[name] = (rs.standard_normal(shape) * g * (1.0 / (cin * k)) ** 0.5).astype(np.float32)
    return sd


def fold_weight_norm(sd: Dict[str, "torch.Tensor | np.ndarray"]) -> Dict[str, np.ndarray]:
    """Checkpoint form -> plain weights: w = g * v / ||v|| with the norm over all dims but 0, what
    torch.nn.utils.remove_weight_norm leaves in ``.weight`` (Synthesiser.__init__, __init__.py:31-33;
    weight_norm's default dim=0 also for ConvTranspose1d)."""
    out = {}
    for k, t in sd.items():
        a = t.detach().cpu().float().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float32)
        if k.endswith(".weight_g"):
            v = sd[k[:-2] + "_v"]
            v = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)
            norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
            out[k[:-9] + ".weight"] = (a.astype(np.float64) * v / norm).astype(np.float32)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = a
    return out


class Fs2VocConfigC(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("dtype", C.c_int32), ("n_mels", C.c_int32),
                ("initial_channel", C.c_int32), ("n_stages", C.c_int32),
                ("up_rates", C.c_int32 * 8), ("up_kernels", C.c_int32 * 8),
                ("n_kernels", C.c_int32), ("rb_kernels", C.c_int32 * 4), ("rb_dilations", (C.c_int32 * 3) * 4)]


def _config_to_c(cfg: HifiGanConfig, dtype: int) -> Fs2VocConfigC:
    c = Fs2VocConfigC()
    c.abi_version, c.dtype, c.n_mels = _lib.FS2_ABI_VERSION, dtype, cfg.num_mels
    c.initial_channel, c.n_stages = cfg.upsample_initial_channel, len(cfg.upsample_rates)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        c.up_rates[i], c.up_kernels[i] = u, k
    c.n_kernels = len(cfg.resblock_kernel_sizes)
    for j, (k, ds) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
        c.rb_kernels[j] = k
        if len(ds) != 3:
            raise ValueError("resblock '1' has three dilations per kernel size")
        for m, d in enumerate(ds):
            c.rb_dilations[j][m] = d
    return c


def _bind(lib):
    vp, i32 = C.c_void_p, C.c_int32
    lib.fs2_voc_create.argtypes = [C.POINTER(Fs2VocConfigC), C.POINTER(vp)]
    lib.fs2_voc_destroy.argtypes = [vp]
    lib.fs2_voc_last_error.restype = C.c_char_p
    lib.fs2_voc_last_error.argtypes = [vp]
    lib.fs2_voc_load_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]
    lib.fs2_voc_finalize.argtypes = [vp]
    lib.fs2_voc_hop.argtypes = [vp]
    lib.fs2_voc_hop.restype = i32
    lib.fs2_voc_synthesize.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.fs2_voc_debug_copy.argtypes = [vp, i32, vp, vp]
    return lib


class HifiGan:
    """Device-resident generator.  ``state_dict`` may be the checkpoint form (weight_g / weight_v) or
    plain weights; names are the reference generator's own."""

    def __init__(self, cfg: HifiGanConfig, state_dict, precision: str = "bf16", device="cuda:0"):
        self.cfg, self.device = cfg, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HiFi-GAN generator runs on an MI355X only (no CPU fallback)")
        self.lib = _bind(_lib.load())
        self.dtype = {"bf16": _lib.FS2_BF16, "fp32": _lib.FS2_F32}[precision]
        self.handle = C.c_void_p()
        cc = _config_to_c(cfg, self.dtype)
        with torch.cuda.device(self.device):  # no global set_device: the reference Synthesiser has no such side effect
            self._init(cfg, cc, state_dict)
        self._last = None

    def _init(self, cfg, cc, state_dict):
        st = self.lib.fs2_voc_create(C.byref(cc), C.byref(self.handle))
        if st != 0:
            raise RuntimeError(f"fs2_voc_create: {_lib.load().fs2_status_string(st).decode()}")
        sd = fold_weight_norm(state_dict)
        spec = state_dict_spec(cfg)
        missing = [k for k in spec if k not in sd]
        if missing:
            raise KeyError(f"generator weights missing: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for name in spec:
            a = np.ascontiguousarray(sd[name], dtype=np.float32)
            shp = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.fs2_voc_load_weight(self.handle, name.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim),
                        f"load_weight {name}")
        self._check(self.lib.fs2_voc_finalize(self.handle), "finalize")
        self.hop = int(self.lib.fs2_voc_hop(self.handle))

    def _check(self, status, what):
        if status != 0:
            raise RuntimeError(f"{what}: {_lib.load().fs2_status_string(status).decode()}: "
                               f"{self.lib.fs2_voc_last_error(self.handle).decode()}")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.fs2_voc_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def synthesize(self, mel: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mel (B, T, n_mels) fp32 -> wav (B, T*hop) fp32 in [-1, 1]; with ``lengths`` (B,) every
        utterance is synthesised from its first lengths[b] frames only (samples past it are 0)."""
        mel = mel.to(self.device, torch.float32).contiguous()
        B, T, M = mel.shape
        if M != self.cfg.num_mels:
            raise ValueError(f"mel has {M} bins, generator wants {self.cfg.num_mels}")
        wav = torch.zeros(B, T * self.hop, dtype=torch.float32, device=self.device)
        ld = None if lengths is None else lengths.to(self.device, torch.int32).contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):  # arena hipMalloc + launches must hit the engine's device
            self._check(self.lib.fs2_voc_synthesize(self.handle, C.c_void_p(mel.data_ptr()),
                                                    None if ld is None else C.c_void_p(ld.data_ptr()), B, T,
                                                    C.c_void_p(wav.data_ptr()), stream), "synthesize")
        self._last = (B, T)
        return wav

    def debug_stage(self, stage: int) -> torch.Tensor:
        B, T = self._last
        up = int(np.prod(self.cfg.upsample_rates[:stage])) if stage else 1
        t = torch.empty(B, T * up, self.cfg.channels()[stage], dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            self._check(self.lib.fs2_voc_debug_copy(self.handle, stage, C.c_void_p(t.data_ptr()), stream), "debug_copy")
        return t


class Synthesiser:
    """Drop-in for ``litfass.third_party.hifigan.Synthesiser`` (__init__.py:19-43): ``synth(mel)`` with
    mel ``(T, num_mels)`` returns int16 numpy ``(1, T*hop)``.  ``checkpoint`` is the path of a
    ``generator_*.pth.tar`` (``{"generator": state_dict}``) or a state_dict."""

    def __init__(self, device="cuda:0", model="universal", checkpoint=None, config: Optional[HifiGanConfig] = None,
                 precision: str = "bf16"):
        self.device = device
        cfg = config or HifiGanConfig()
        if checkpoint is None:
            raise FileNotFoundError(f"generator_{model}.pth.tar is not shipped with the reference repository "
                                    "(a Git-LFS blob); pass checkpoint=<path or state_dict>")
        if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
            checkpoint = torch.load(checkpoint, map_location="cpu")
        sd = checkpoint.get("generator", checkpoint)
        self.vocoder = HifiGan(cfg, sd, precision=precision, device=device)

    def __call__(self, mel):
        mel = torch.as_tensor(mel, dtype=torch.float32)
        wav = self.vocoder.synthesize(mel.unsqueeze(0))  # (1, T*hop)
        return (wav.cpu().numpy() * 32768.0).astype("int16")
