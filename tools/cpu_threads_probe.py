import sys, time, math
sys.path.insert(0, '.')
import torch
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
from oracle import oracle_cpu
cfg = preset('c2')
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
inp = synth_inputs(cfg, 32, 256, seed=1234)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    B = 8
    oracle_cpu.forward(sd, cfg, inp['phones'][:1], inp['speaker'][:1])
    t0 = time.perf_counter(); out = oracle_cpu.forward(sd, cfg, inp['phones'][:B], inp['speaker'][:B]); t = time.perf_counter() - t0
    print(nt, 'threads', round(B * 1536 / t), 'frames/s', round(t, 2), 's', flush=True)
