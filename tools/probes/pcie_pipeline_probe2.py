"""Repeatability of the host-boundary pipeline: copies on the replica's copy stream vs on the forward's own stream, 3 alternations."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
cfg = preset("c2")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
model = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
host = {"phones": torch.from_numpy(inp["phones"]).pin_memory(), "speaker": torch.from_numpy(inp["speaker"]).pin_memory()}
def run(n, label, N=40):
    pipe = model.pipeline(n, host_outputs=("mel", "tgt_mask"))
    for _ in range(4 * n + 1):
        pipe.submit(host)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        pipe.submit(host)
    pipe.drain(); torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) / N * 1e3:.3f} ms/batch", flush=True)
    pipe.close()
for rep in range(3):
    for mode in ("own", "same"):
        os.environ["FS2_PIPE_COPY_STREAM"] = mode
        for n in (2, 3):
            run(n, f"rep {rep} copy stream {mode}, {n} in flight")
