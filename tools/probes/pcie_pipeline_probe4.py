"""Which host call stalls in the host-boundary pipeline?  Wraps ForwardPipeline._to_host / _run pieces with timers and prints the slowest."""
import gc, math, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd import model as M
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
cfg = preset("c2")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
model = M.FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
host = {"phones": torch.from_numpy(inp["phones"]).pin_memory(), "speaker": torch.from_numpy(inp["speaker"]).pin_memory()}
rec = {"to_host": [], "forward": [], "handover": []}
lock = threading.Lock()
orig_to_host = M.ForwardPipeline._to_host
def timed_to_host(self, k, out, done):
    t0 = time.perf_counter(); r = orig_to_host(self, k, out, done); dt = time.perf_counter() - t0
    with lock: rec["to_host"].append(dt * 1e3)
    return r
M.ForwardPipeline._to_host = timed_to_host
orig_call = M.FastSpeech2.__call__
def timed_call(self, *a, **kw):
    t0 = time.perf_counter(); r = orig_call(self, *a, **kw); dt = time.perf_counter() - t0
    with lock: rec["forward"].append(dt * 1e3)
    return r
M.FastSpeech2.__call__ = timed_call
orig_ho = M.ForwardPipeline._hand_over
def timed_ho(self, fut):
    t0 = time.perf_counter(); r = orig_ho(self, fut); dt = time.perf_counter() - t0
    rec["handover"].append(dt * 1e3)
    return r
M.ForwardPipeline._hand_over = timed_ho
for rep in range(6):
    pipe = model.pipeline(2, host_outputs=("mel", "tgt_mask"))
    for _ in range(9): pipe.submit(host)
    pipe.drain(); torch.cuda.synchronize(); gc.collect(); gc.disable()
    for v in rec.values(): v.clear()
    ts = []; t0 = time.perf_counter()
    for _ in range(40):
        pipe.submit(host); ts.append(time.perf_counter())
    pipe.drain(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    d = np.diff(np.array([t0] + ts)) * 1e3
    print(f"rep {rep}: {el / 40 * 1e3:.3f} ms/batch; submit max {d.max():.1f} ms; forward host ms: med {np.median(rec['forward']):.2f} max {max(rec['forward']):.1f}; "
          f"to_host ms: med {np.median(rec['to_host']):.2f} max {max(rec['to_host']):.1f}; hand-over ms: med {np.median(rec['handover']):.2f} max {max(rec['handover']):.1f}", flush=True)
    pipe.close(); gc.enable()
