"""fwd_loop.py with the per-class profiling events of bench.py's timed region switched on / off."""
import sys, time, math, torch
sys.path[:0] = ["."]
from lightningfastspeech2_amd import _lib
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
cfg = preset("c2")
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
model = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
inp = synth_inputs(cfg, 32, 256, seed=1234)
batch = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}
for _ in range(5): out = model(batch, inference=True)
for prof in (0, 1, 0, 1):
    if prof:
        model.engine.profile_reserve(_lib.K_DEC_FFN_CONV1, 400)
        model.engine.profile_enable(_lib.K_DEC_FFN_CONV1, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = model(batch, inference=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    if prof:
        r = model.engine.profile_read(_lib.K_DEC_FFN_CONV1)
        model.engine.profile_enable(_lib.K_DEC_FFN_CONV1, False)
    print("profiling", prof, f"{dt*1e3:.3f} ms per forward")
