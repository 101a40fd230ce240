"""Does splitting the batch over two streams (two engines, two host threads) beat one engine on the whole batch?  The encoder's
launches are latency-bound (128-256 workgroups, 10-20 us each), and utterances are independent."""
import sys, time, math, threading, torch
sys.path[:0] = ["."]
from lightningfastspeech2_amd.config import preset
from lightningfastspeech2_amd.model import FastSpeech2
from lightningfastspeech2_amd.weights import synth_inputs, synth_state_dict
cfg = preset(sys.argv[1] if len(sys.argv) > 1 else "c2")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sd = synth_state_dict(cfg, 0, duration_bias=math.log(7.0), duration_weight_scale=0.0)
inp = synth_inputs(cfg, B, 256, seed=1234)
ph, sp = torch.from_numpy(inp["phones"]).cuda(), torch.from_numpy(inp["speaker"]).cuda()
N = 30


def run(models, parts, streams):
    def work(m, b, s):
        with torch.cuda.stream(s):
            for _ in range(N): m(b, inference=True)
    for m, b, s in zip(models, parts, streams):  # warm-up
        with torch.cuda.stream(s):
            for _ in range(5): m(b, inference=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=a) for a in zip(models, parts, streams)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


one = FastSpeech2(cfg, sd, precision="bf16", device="cuda:0")
print(f"one engine, B = {B}: {run([one], [{'phones': ph, 'speaker': sp}], [torch.cuda.Stream()]):.3f} ms per batch")
for k in (2, 4):
    ms = [FastSpeech2(cfg, sd, precision="bf16", device="cuda:0") for _ in range(k)]
    h = B // k
    parts = [{"phones": ph[i * h:(i + 1) * h].contiguous(), "speaker": sp[i * h:(i + 1) * h].contiguous()} for i in range(k)]
    print(f"{k} engines x B = {h} on {k} streams: {run(ms, parts, [torch.cuda.Stream() for _ in range(k)]):.3f} ms per batch")
