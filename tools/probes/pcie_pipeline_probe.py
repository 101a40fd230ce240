"""Where does the host boundary inside ForwardPipeline (host_outputs) lose time?  Variants: device-only pipeline; host inputs only;
host outputs on a copy stream queued by a copier thread (the shipped form); a bare D2H loop beside a running
pipeline.  Prints ms per batch for each."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2, ForwardPipeline
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict

cfg = preset(os.environ.get("CFG", "c2"))
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
model = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
ph, sp = torch.from_numpy(inp["phones"]).pin_memory(), torch.from_numpy(inp["speaker"]).pin_memory()
dev = {"phones": ph.cuda(), "speaker": sp.cuda()}
host = {"phones": ph, "speaker": sp}
N = 30


def run(pipe, batch, label):
    for _ in range(10):
        pipe.submit(batch)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        pipe.submit(batch)
    pipe.drain(); torch.cuda.synchronize()
    print(f"{label:70s} {(time.perf_counter() - t0) / N * 1e3:.3f} ms/batch", flush=True)
    pipe.close()


run(model.pipeline(2), dev, "device in / device out, 2 in flight")
run(model.pipeline(2, host_outputs=("src_mask",)), host, "pinned host in / only the 8 KB src_mask out, 2 in flight")
run(model.pipeline(2, host_outputs=("mel", "tgt_mask")), host, "host in / host out, copies on a copy stream, 2 in flight")
run(model.pipeline(3, host_outputs=("mel", "tgt_mask")), host, "host in / host out, copies on a copy stream, 3 in flight")
run(model.pipeline(2, host_outputs=("mel",)), host, "host in / host mel only, copy stream, 2 in flight")
# a bare D2H of 15.7 MB alone and beside the pipeline
mel = torch.empty(32, 1536, 80, device="cuda:0")
hm = torch.empty(32, 1536, 80).pin_memory()
cs = torch.cuda.Stream()
def d2h_loop(n):
    with torch.cuda.stream(cs):
        t0 = time.perf_counter()
        for _ in range(n):
            hm.copy_(mel, non_blocking=True)
        cs.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
d2h_loop(3)
print(f"bare D2H of 15.7 MB, idle GPU: {d2h_loop(20):.3f} ms each", flush=True)
pipe = model.pipeline(2)
for _ in range(6):
    pipe.submit(dev)
import threading
res = {}
th = threading.Thread(target=lambda: res.setdefault("t", d2h_loop(40)))
th.start()
t0 = time.perf_counter(); n = 0
while th.is_alive():
    pipe.submit(dev); n += 1
pipe.drain(); torch.cuda.synchronize()
print(f"bare D2H beside a 2-in-flight pipeline: {res['t']:.3f} ms each; the pipeline meanwhile {(time.perf_counter() - t0) / max(n, 1) * 1e3:.3f} ms/batch", flush=True)
pipe.close()
