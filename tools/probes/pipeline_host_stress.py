"""Soak test of ForwardPipeline(host_outputs=...): 900 ragged batches of changing shapes through 2 / 3 / 4 forwards in flight, every host
output compared bit for bit with the synchronous forward (ring-slot reuse, copier threads, upload streams).  r06: 0 mismatches."""
import numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_boundary import _case, _model
from lightningfastspeech2_amd.weights import synth_inputs
cfg, sd, inp, batch = _case()
m = _model(cfg, sd, "bf16")
rs = np.random.RandomState(9)
batches = []
for i in range(60):
    B = int(rs.randint(1, 6)); L = int(rs.randint(5, 40))
    lens = sorted((int(rs.randint(1, L + 1)) for _ in range(B)), reverse=True); lens[0] = L
    x = synth_inputs(cfg, B, L, seed=500 + i, lengths=lens)
    batches.append({"phones": torch.from_numpy(x["phones"]).pin_memory(), "speaker": torch.from_numpy(x["speaker"]).pin_memory()})
want = [{k: v.cpu() for k, v in m(b, inference=True).items()} for b in batches]
bad = 0
for n in (2, 3, 4):
    pipe = m.pipeline(n, host_outputs=("mel", "tgt_mask"))
    got = []
    def take(outs):
        for o in outs:
            got.append({k: v.cpu().clone() for k, v in o.items()})
    for rep in range(5):
        for b in batches: take(pipe.submit(b))
    take(pipe.drain()); pipe.close()
    assert len(got) == 5 * len(want)
    for i, g in enumerate(got):
        w = want[i % len(want)]
        for k in w:
            if not torch.equal(g[k], w[k]): bad += 1
    print("in flight", n, "results", len(got), "mismatches so far", bad, flush=True)
assert bad == 0
print("stress ok")
