"""MI355X-native FastSpeech2 / LightSpeech mel forward (drop-in for litfass.fastspeech2 forward)."""
from .config import Fs2Config, preset  # noqa: F401
from .weights import state_dict_spec, synth_state_dict, synth_inputs  # noqa: F401

__all__ = ["Fs2Config", "preset", "state_dict_spec", "synth_state_dict", "synth_inputs", "FastSpeech2", "Trainer"]


def __getattr__(name):
    # the engine needs the HIP library; keep config/weights importable without it
    if name == "FastSpeech2":
        from .model import FastSpeech2
        return FastSpeech2
    if name == "Trainer":  # the training step: forward tape + backward + clip + AdamW / Noam
        from .training import Trainer
        return Trainer
    raise AttributeError(name)
