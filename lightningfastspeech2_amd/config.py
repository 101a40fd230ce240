"""Shape/hyper-parameter description of the FastSpeech2 / LightSpeech mel forward.

Field names follow the reference's ``FastSpeech2.__init__`` keyword arguments
(/root/reference/litfass/fastspeech2/fastspeech2.py:46-130) so a Lightning
``hparams`` namespace / dict can be converted with :meth:`Fs2Config.from_hparams`.
Only the hparams that decide the shapes and arithmetic of ``FastSpeech2.forward``
(fastspeech2.py:636-731) are kept; training/logging flags are not part of the path.
"""
from __future__ import annotations

import copy
import json
from dataclasses import asdict, dataclass, field
from typing import Dict, List

DVECTOR_DIM = 256  # SpeakerEmbedding.projection = nn.Linear(256, H)   (model.py:131)
PE_MAX_LEN = 5000  # PositionalEncoding(max_len=5000)                  (model.py:39)


def _default_stats(variances):
    return {v: {"min": -3.0, "max": 3.0, "mean": 0.0, "std": 1.0} for v in variances}


@dataclass
class Fs2Config:
    # phone table size: len(phone2id)                                   (fastspeech2.py:242-245)
    n_phones: int = 80
    encoder_hidden: int = 256
    encoder_head: int = 2
    encoder_layers: int = 4
    encoder_kernel_sizes: List[int] = field(default_factory=lambda: [5, 25, 13, 9])
    encoder_depthwise_conv: bool = True
    encoder_conv_filter_size: int = 1024
    decoder_hidden: int = 256
    decoder_head: int = 2
    decoder_layers: int = 4
    decoder_kernel_sizes: List[int] = field(default_factory=lambda: [17, 21, 9, 13])
    decoder_depthwise_conv: bool = True
    decoder_conv_filter_size: int = 1024
    variances: List[str] = field(default_factory=lambda: ["pitch", "energy", "snr"])
    variance_levels: List[str] = field(default_factory=lambda: ["frame", "frame", "frame"])
    variance_transforms: List[str] = field(default_factory=lambda: ["none", "none", "none"])
    variance_nlayers: List[int] = field(default_factory=lambda: [5, 5, 5])
    variance_kernel_size: List[int] = field(default_factory=lambda: [3, 3, 3])
    variance_filter_size: int = 256
    variance_nbins: int = 256
    variance_depthwise_conv: bool = True
    duration_nlayers: int = 2
    duration_kernel_size: int = 3
    duration_filter_size: int = 256
    duration_depthwise_conv: bool = True
    n_mels: int = 80
    sampling_rate: int = 22050
    hop_length: int = 256
    max_length: float = 32.0  # seconds                                  (fastspeech2.py:55)
    speaker_type: str = "dvector"
    priors: List[str] = field(default_factory=list)
    stats: Dict[str, Dict[str, float]] = field(default_factory=dict)

    def __post_init__(self):
        if not self.stats:
            self.stats = _default_stats(self.variances)
        self.validate()

    # ---- derived ---------------------------------------------------------------------------
    @property
    def hidden(self) -> int:
        return self.encoder_hidden

    @property
    def max_frames(self) -> int:
        # VarianceAdaptor(max_length = max_length*sr/hop) then int() in LengthRegulator
        # (fastspeech2.py:341-343, model.py:355): 32*22050/256 = 2756.25 -> 2756
        return int(self.max_length * self.sampling_rate / self.hop_length)

    def validate(self):
        H = self.encoder_hidden
        if self.decoder_hidden != H:
            raise ValueError("decoder_hidden must equal encoder_hidden: the variance adaptor "
                             "output feeds the decoder unchanged (fastspeech2.py:703-721)")
        if self.speaker_type != "dvector":
            # speaker_type='id' crashes in the reference (model.py:130-138)
            raise ValueError("only speaker_type='dvector' is a working reference path")
        for name, heads in (("encoder", self.encoder_head), ("decoder", self.decoder_head)):
            if H % heads:
                raise ValueError(f"{name}_head must divide hidden")
        if len(self.encoder_kernel_sizes) < self.encoder_layers:
            raise ValueError("encoder_kernel_sizes shorter than encoder_layers")
        if len(self.decoder_kernel_sizes) < self.decoder_layers:
            raise ValueError("decoder_kernel_sizes shorter than decoder_layers")
        for dw, F in ((self.encoder_depthwise_conv, self.encoder_conv_filter_size),
                      (self.decoder_depthwise_conv, self.decoder_conv_filter_size)):
            if dw and F % H:
                raise ValueError("depth-wise FFN needs conv_filter_size % hidden == 0 "
                                 "(grouped conv2.0, model.py:84-91)")
        nv = len(self.variances)
        if not (len(self.variance_levels) >= nv and len(self.variance_transforms) >= nv
                and len(self.variance_nlayers) >= nv and len(self.variance_kernel_size) >= nv):
            raise ValueError("variance_* lists shorter than variances")
        for i, v in enumerate(self.variances):
            if self.variance_levels[i] not in ("frame", "phone"):  # model.py:276 / :316
                raise ValueError(f"variance_levels[{i}]={self.variance_levels[i]!r}: the reference knows 'frame' and 'phone'")
            if self.variance_transforms[i] not in ("none", "cwt"):
                raise ValueError(f"variance_transforms[{i}]={self.variance_transforms[i]!r}: the reference knows 'none' and 'cwt'")
            if self.variance_transforms[i] == "cwt":  # VarianceEncoder takes log(min), log(max) (model.py:394-396)
                st = self.stats.get(v, {})
                if not (st.get("min", 0) > 0 and st.get("max", 0) > st.get("min", 0)):
                    raise ValueError(f"variance {v!r} uses the CWT head: stats min/max must be positive (bins are log-spaced)")
            if self.variance_nlayers[i] > 1 and self.variance_filter_size != H:
                raise ValueError("variance_filter_size must equal hidden when nlayers>1 "
                                 "(every layer is built in_channels->filter, model.py:497-501)")
            if v not in self.stats:
                raise ValueError(f"stats missing for variance {v!r}")
        if self.duration_nlayers > 1 and self.duration_filter_size != H:
            raise ValueError("duration_filter_size must equal hidden when nlayers>1")
        for pr in self.priors:
            # PriorEmbedding(H, variance_nbins, stats[f"{prior}_prior"]) (fastspeech2.py:416-424)
            st = self.stats.get(f"{pr}_prior")
            if not st or "min" not in st or "max" not in st:
                raise ValueError(f"stats['{pr}_prior'] with min/max is required for prior {pr!r}")

    # ---- (de)serialisation ----------------------------------------------------------------
    def is_cwt(self, i: int) -> bool:
        return self.variance_transforms[i] == "cwt"

    def is_phone_level(self, i: int) -> bool:
        """variance i is predicted per phone, before the length regulator (model.py:276-294), not per frame (:315-333)"""
        return self.variance_levels[i] == "phone"

    def to_dict(self) -> dict:
        return asdict(self)

    def to_json(self) -> str:
        return json.dumps(self.to_dict(), sort_keys=True)

    @classmethod
    def from_dict(cls, d: dict) -> "Fs2Config":
        known = {f for f in cls.__dataclass_fields__}
        return cls(**{k: copy.deepcopy(v) for k, v in d.items() if k in known})

    @classmethod
    def from_json(cls, s: str) -> "Fs2Config":
        return cls.from_dict(json.loads(s))

    @classmethod
    def from_hparams(cls, hparams, stats: dict, n_phones: int) -> "Fs2Config":
        """Build from a reference ``hparams`` object/dict (save_hyperparameters,
        fastspeech2.py:150-163) plus the checkpoint extras ``stats`` / ``len(phone2id)``."""
        get = (lambda k, d=None: hparams.get(k, d)) if isinstance(hparams, dict) else \
              (lambda k, d=None: getattr(hparams, k, d))
        kw = {}
        for f in cls.__dataclass_fields__:
            if f in ("n_phones", "stats"):
                continue
            v = get(f)
            if v is not None:
                kw[f] = copy.deepcopy(v)
        nv = len(kw.get("variances", cls().variances))
        for f in ("variance_levels", "variance_transforms", "variance_nlayers", "variance_kernel_size"):
            if f in kw:
                kw[f] = list(kw[f])[:nv]
        return cls(n_phones=n_phones, stats=copy.deepcopy(stats), **kw)

    # ---- parameter count (matches sum(p.numel()) of the reference modules) ----------------
    def param_count(self) -> int:
        from .weights import state_dict_spec
        return sum(int(_prod(s)) for n, s in state_dict_spec(self).items()
                   if not n.endswith(".pe") and not n.endswith(".bins"))


def _prod(shape):
    p = 1
    for s in shape:
        p *= s
    return p


# ---- the configurations BASELINE.json / SURVEY.md §8(d) name --------------------------------
def preset(name: str) -> Fs2Config:
    name = name.lower()
    v3 = dict(variances=["pitch", "energy", "snr"], variance_nlayers=[5, 5, 5],
              variance_kernel_size=[3, 3, 3])
    if name in ("c1", "c2", "fs2-27m"):
        # FS2-27M: dense convs k=9, H=256, F=1024, heads 2, 4+4 layers -> 26.76 M params
        return Fs2Config(encoder_hidden=256, decoder_hidden=256, encoder_head=2, decoder_head=2,
                         encoder_layers=4, decoder_layers=4,
                         encoder_kernel_sizes=[9] * 4, decoder_kernel_sizes=[9] * 4,
                         encoder_depthwise_conv=False, decoder_depthwise_conv=False,
                         encoder_conv_filter_size=1024, decoder_conv_filter_size=1024,
                         variance_filter_size=256, variance_depthwise_conv=False,
                         duration_filter_size=256, duration_depthwise_conv=False, **v3)
    if name in ("c3", "c4", "ls-76m"):
        # LS-76M: all depth-wise, H=768, F=3072, heads 6, enc 4 / dec 5 layers -> 75.06 M
        return Fs2Config(encoder_hidden=768, decoder_hidden=768, encoder_head=6, decoder_head=6,
                         encoder_layers=4, decoder_layers=5,
                         encoder_kernel_sizes=[5, 25, 13, 9], decoder_kernel_sizes=[17, 21, 9, 13, 9],
                         encoder_depthwise_conv=True, decoder_depthwise_conv=True,
                         encoder_conv_filter_size=3072, decoder_conv_filter_size=3072,
                         variance_filter_size=768, variance_depthwise_conv=True,
                         duration_filter_size=768, duration_depthwise_conv=True, **v3)
    if name in ("c5", "fs2-1b"):
        # FS2-1B: dense, H=1024, F=4096, heads 8, 12+12 layers k=9 -> 1.162 B
        return Fs2Config(encoder_hidden=1024, decoder_hidden=1024, encoder_head=8, decoder_head=8,
                         encoder_layers=12, decoder_layers=12,
                         encoder_kernel_sizes=[9] * 12, decoder_kernel_sizes=[9] * 12,
                         encoder_depthwise_conv=False, decoder_depthwise_conv=False,
                         encoder_conv_filter_size=4096, decoder_conv_filter_size=4096,
                         variance_filter_size=1024, variance_depthwise_conv=False,
                         duration_filter_size=1024, duration_depthwise_conv=False, **v3)
    if name in ("ref-default", "ls-7.8m"):
        # the reference's own defaults (depth-wise, H=256, 4+4) -> 7.84 M; sanity config
        return Fs2Config(**v3)
    raise KeyError(name)
