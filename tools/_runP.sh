python -m pytest tests/test_gpu_forward.py tests/test_gpu_boundary.py tests/test_gpu_parity_corners.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import math, time, torch, sys
sys.path.insert(0, '.')
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
for name in ("c2", "ref-default"):
    cfg = preset(name)
    for fpp in (2, 3):   # frames per phone -> T = 512 / 768
        sd = synth_state_dict(cfg, 0, duration_bias=math.log(1.0 + fpp), duration_weight_scale=0.0)
        m = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
        inp = synth_inputs(cfg, 32, 256, seed=1234)
        batch = {"phones": torch.from_numpy(inp["phones"]).cuda(), "speaker": torch.from_numpy(inp["speaker"]).cuda()}
        res = {}
        for knob in (1340, 1341, 1340, 1341):
            m.engine.set_tuning(knob)
            for _ in range(3): out = m(batch, inference=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): out = m(batch, inference=True)
            torch.cuda.synchronize(); res.setdefault(knob, []).append((time.perf_counter() - t0) / 20 * 1e3)
        print(name, "T =", out["mel"].shape[1], "two launches", [round(v, 3) for v in res[1340]], "one launch", [round(v, 3) for v in res[1341]], flush=True)
PY
