#!/usr/bin/env python
"""Probe: does the HiFi-GAN pass gain from n utterance groups on n HIP streams (one generator replica + host thread each)?
    python tools/probes/voc_streams_probe.py [--batch 32 --frames 1536]"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lightningfastspeech2_amd.hifigan import HifiGan, HifiGanConfig, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1536)
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    cfg = HifiGanConfig()
    sd = synth_state_dict(cfg, 0)
    rs = np.random.RandomState(1234)
    mel = torch.from_numpy((rs.standard_normal((a.batch, a.frames, 80)) * 1.5 - 4.0).astype(np.float32)).cuda()
    for n, full in ((1, 0), (2, 0), (4, 0), (2, 1)):
        bg = a.batch if full else a.batch // n
        gens = [HifiGan(cfg, sd, precision="bf16") for _ in range(n)]
        streams = [torch.cuda.Stream() for _ in range(n)]
        mels = [mel if full else mel[i * bg:(i + 1) * bg].contiguous() for i in range(n)]

        def work(i, k):
            with torch.cuda.stream(streams[i]):
                for _ in range(k):
                    gens[i].synthesize(mels[i])
                streams[i].synchronize()

        def run(k):
            ths = [threading.Thread(target=work, args=(i, k)) for i in range(n)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k * 1e3 / (n if full else 1)
        run(2)
        print(f"{n} stream(s), {'full batches' if full else 'groups'} of {bg}: {min(run(a.steps), run(a.steps)):.2f} ms per batch-{a.batch} pass", flush=True)
        del gens


if __name__ == "__main__":
    main()
