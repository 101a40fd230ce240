"""The boundary takes device pointers; this measures what a caller pays if its batch starts and its mels
end in (pinned) host memory: H2D of phones + speaker, forward, D2H of mel + masks.  Not bench.py's value."""
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
ph, sp = torch.from_numpy(inp["phones"]).pin_memory(), torch.from_numpy(inp["speaker"]).pin_memory()
mel_h = None
def step(transfers):
    global mel_h
    batch = {"phones": ph.cuda(non_blocking=True), "speaker": sp.cuda(non_blocking=True)} if transfers else step.dev
    out = model(batch, inference=True)
    if transfers:
        if mel_h is None:
            mel_h = torch.empty(out["mel"].shape, dtype=torch.float32).pin_memory()
        mel_h.copy_(out["mel"], non_blocking=True)
        out["tgt_mask"].cpu()
    return out
step.dev = {"phones": ph.cuda(), "speaker": sp.cuda()}
for tr in (False, True):
    for _ in range(3): step(tr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = step(tr)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    fr = int((~out["tgt_mask"]).sum())
    print(f"transfers={tr}: {dt*1e3:.3f} ms/step, {fr/dt/1e6:.2f} M mel-frames/s")
