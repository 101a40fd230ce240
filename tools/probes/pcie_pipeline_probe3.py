"""Per-step hand-over times of the host-boundary pipeline: uniform slowness or a few stalls?"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
cfg = preset("c2")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
model = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
host = {"phones": torch.from_numpy(inp["phones"]).pin_memory(), "speaker": torch.from_numpy(inp["speaker"]).pin_memory()}
for rep in range(6):
    pipe = model.pipeline(2, host_outputs=("mel", "tgt_mask"))
    for _ in range(9):
        pipe.submit(host)
    pipe.drain(); torch.cuda.synchronize()
    if os.environ.get("GC", "1") == "1":
        import gc
        gc.collect(); torch.cuda.synchronize(); gc.disable()  # (a collected engine replica = fs2_destroy = hipFree = a device-wide stall)
    ts = []
    t0 = time.perf_counter()
    for _ in range(40):
        pipe.submit(host); ts.append(time.perf_counter())
    pipe.drain(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = np.diff(np.array([t0] + ts)) * 1e3
    st = torch.cuda.memory_stats()
    print(f"rep {rep}: {el / 40 * 1e3:.3f} ms/batch; submit-to-submit ms: median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f}; "
          f"device allocs so far {st.get('num_device_alloc', 0)}, reserved {st.get('reserved_bytes.all.current', 0) / 1e6:.0f} MB", flush=True)
    pipe.close()
