#!/usr/bin/env python
"""End-to-end text-to-waveform throughput on one MI355X: FastSpeech2 mel forward (bench.py's workload: FS2-27M,
batch 32 x 256 phonemes, 6 frames/phoneme -> 1536 frames each) followed by the HiFi-GAN V1 generator on the
padded mel batch with its valid frame counts (SpeechGenerator.generate_samples without the host loop), bf16.
Prints ONE JSON line.  Random-init weights, synthetic inputs, everything resident in HBM."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.hifigan import HifiGan, HifiGanConfig, synth_state_dict as voc_sd
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--phones", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    cfg = preset("c2")
    model = FastSpeech2(cfg, synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0), precision="bf16")
    vcfg = HifiGanConfig()
    voc = HifiGan(vcfg, voc_sd(vcfg, 0), precision="bf16")
    inp = synth_inputs(cfg, a.batch, a.phones, seed=1234)
    batch = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}

    def step():
        out = model(batch, inference=True)
        lengths = (~out["tgt_mask"]).sum(dim=1).to(torch.int32)
        return out, voc.synthesize(out["mel"], lengths)

    for _ in range(a.warmup):
        out, wav = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, wav = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.steps
    frames = int((~out["tgt_mask"]).sum())
    samples = frames * vcfg.hop
    print(json.dumps({"metric": "audio samples/sec, phonemes -> waveform (FastSpeech2 FS2-27M + HiFi-GAN V1)", "value": samples / el,
                      "unit": "samples/s", "ms_per_step": el * 1e3, "audio_seconds_per_step": samples / vcfg.sampling_rate,
                      "rtf": el / (samples / vcfg.sampling_rate), "mel_frames_per_s": frames / el, "n_gpus": 1, "dtype": "bf16",
                      "data": "synthetic", "steps": a.steps, "warmup": a.warmup,
                      "config": {"workload": f"batch {a.batch} x {a.phones} phonemes -> {frames // a.batch} frames -> "
                                             f"{samples // a.batch} samples per utterance, random-init weights"}}), flush=True)


if __name__ == "__main__":
    main()
