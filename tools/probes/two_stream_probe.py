#!/usr/bin/env python
"""Principle probe: does the C2 forward gain from running utterance groups on concurrent HIP streams?
N threads x (own engine, own stream, B = 32 / N utterances) against one engine with B = 32.  ctypes releases the GIL during the
library calls, so the threads drive their streams concurrently (each forward has its own host sync).
    python tools/probes/two_stream_probe.py [--config c2] [--steps 30]
"""
import argparse
import math
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lightningfastspeech2_amd.config import preset  # noqa: E402
from lightningfastspeech2_amd.model import FastSpeech2  # noqa: E402
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--graphs", type=int, default=0)
    ap.add_argument("--full", type=int, default=0, help="1: every stream runs FULL batches (n batches in flight: does hiding one forward's host-sync seam under another's kernels pay?)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = preset(a.config)
    sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
    inp = synth_inputs(cfg, a.batch, 256, seed=1234)
    phones, spk = torch.from_numpy(inp["phones"]).to(dev), torch.from_numpy(inp["speaker"]).to(dev)
    for n in (1, 2, 4):
        if a.batch % n:
            continue
        bg = a.batch if a.full else a.batch // n
        models = [FastSpeech2(cfg, sd, precision="bf16", device=dev) for _ in range(n)]
        streams = [torch.cuda.Stream(dev) for _ in range(n)]
        batches = [{"phones": phones[i * bg:(i + 1) * bg].contiguous(), "speaker": spk[i * bg:(i + 1) * bg].contiguous()} for i in range(n)]
        if a.full:
            batches = [{"phones": phones, "speaker": spk} for _ in range(n)]
        for m in models:
            m.engine.set_graphs(bool(a.graphs))

        def work(i, k):
            with torch.cuda.stream(streams[i]):
                for _ in range(k):
                    models[i](batches[i], inference=True)
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
            return (time.perf_counter() - t0) / k * 1e3 / (n if a.full else 1)
        run(5)
        best = min(run(a.steps) for _ in range(3))
        print(f"{a.config} B={a.batch}: {n} group(s) of {bg} on {n} stream(s): {best:.3f} ms per batch-{a.batch} step (graphs={a.graphs})", flush=True)
        del models


if __name__ == "__main__":
    main()
