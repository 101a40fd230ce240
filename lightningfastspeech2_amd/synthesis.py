"""Mel forward -> vocoder, the core of the reference's ``SpeechGenerator.generate_samples``
(litfass/synthesis/generator.py:151-223) without its optional post-processing (voicefixer, audio
augmentations — off-path).

The reference loops over utterances on the host: ``mel = result["mel"][i][~result["tgt_mask"][i]]``
goes to the CPU and back, one ``Synthesiser`` call each (generator.py:163-170).  Here the padded mel
batch never leaves HBM: the generator takes ``(B, T, 80)`` plus the valid frame counts and
synthesises every utterance from its own frames only (zero padding at ITS ends in every layer), which
is what the per-utterance loop computes.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from .hifigan import HifiGan
from .model import FastSpeech2

INT16_MAX = float(np.iinfo(np.int16).max)


def int16_samples_to_float32(y: np.ndarray) -> np.ndarray:
    """generator.py:24-33."""
    if y.dtype == np.float32:
        return y
    if y.dtype != np.int16:
        raise ValueError(f"input samples not int16 or float32, but {y.dtype}")
    return y.astype(np.float32) / INT16_MAX


class SpeechGenerator:
    """``generate_samples(batch)`` -> ``{"fs", "audios"[, "durations"]}`` like the reference's
    (generator.py:151-223): ``audios`` is a list of float32 arrays, one per utterance, each the
    int16-quantised generator output rescaled by 1/32767 exactly as the reference does
    (Synthesiser.__call__ then int16_samples_to_float32)."""

    def __init__(self, model: FastSpeech2, vocoder: HifiGan, g2p_model=None):
        self.model, self.synth, self.g2p = model, vocoder, g2p_model
        self.model.eval()

    def generate_from_text(self, text: str, speaker) -> np.ndarray:
        """generator.py:96-150 for the default (no priors) d-vector model: text -> phones the model
        knows -> batch of one -> audio.  ``speaker`` is a key of ``model.speaker2dvector`` or a
        256-d vector (the reference draws a random speaker when none is given; here it is explicit)."""
        from .frontend import text_to_batch
        if self.g2p is None:
            raise RuntimeError("no G2P model was given to SpeechGenerator")
        dvec = self.model.speaker2dvector[speaker] if not hasattr(speaker, "__len__") or isinstance(speaker, str) else speaker
        batch = text_to_batch(self.model.phone2id, self.g2p, text, dvec)
        return self.generate_samples(batch)["audios"][0]

    def generate_samples(self, batch: Dict, return_duration: bool = False) -> Dict:
        result = self.model(batch, inference=True)                       # generator.py:158
        lengths = (~result["tgt_mask"]).sum(dim=1).to(torch.int32)       # frames the reference keeps, :163
        wav = self.synth.synthesize(result["mel"], lengths)               # (B, T*hop) fp32, device
        # the float -> int16 cast happens on the host with numpy, exactly as Synthesiser.__call__ does it
        # (__init__.py:39-43): a device-side cast of tanh's +1.0 * 32768 is out of range (undefined)
        i16 = (wav.cpu().numpy() * 32768.0).astype("int16")
        hop = self.synth.hop
        audios: List[np.ndarray] = [int16_samples_to_float32(i16[b, :int(n) * hop]) for b, n in enumerate(lengths.tolist())]
        out = {"fs": self.model.hparams.sampling_rate, "audios": audios}
        if return_duration:
            out["durations"] = [d.cpu() for d in result["duration_rounded"]]
        return out
