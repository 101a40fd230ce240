"""Do all tile heights (MI variants) of the slab GEMM give the same bits for the same rows?  (They must: the
shard == whole property of the forward rests on it.)  python tools/probes/tile_height_invariance.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import _gpu
from lightningfastspeech2_amd import _lib
torch.manual_seed(0)
dt = _gpu.BF16
for (K, taps, name) in ((256, 1, "out_proj+LN"), (1024, 1, "conv2+LN")):
    N = 256
    x = torch.randn(49152, K).cuda(); w = (torch.randn(N, K * taps) / K ** 0.5).cuda()
    b = torch.randn(N).cuda(); res = torch.randn(49152, N).cuda(); g = torch.randn(N).cuda(); be = torch.randn(N).cuda()
    outs = []
    for M in (49152, 12288, 1536, 256):
        y = _gpu.gemm_ln(dt, x[:M].contiguous(), w, b, res[:M].contiguous(), g, be, taps=taps, S=None)
        y = y[0] if isinstance(y, tuple) else y
        outs.append(y[:256].float())
    print(name, [bool(torch.equal(outs[0], o)) for o in outs], float((outs[0]-outs[1]).abs().max()))
    outs = []
    for M in (49152, 12288, 1536, 256):
        y = _gpu.gemm(dt, x[:M].contiguous(), w, b, taps=taps, S=None, relu=False)
        outs.append(y[:256].float())
    print(name, "plain", [bool(torch.equal(outs[0], o)) for o in outs])
# conv shapes with ReLU: decoder conv1 (k=9 -> 1024, plain store) and a predictor layer (k=3, ReLU -> LayerNorm, head)
for (K, N, taps, ln, name) in ((256, 1024, 9, False, "conv1 relu"), (256, 256, 3, True, "predictor layer relu+LN+head")):
    S = 1536
    x = torch.randn(32 * S, K).cuda(); w = (torch.randn(N, K * taps) / (K * taps) ** 0.5).cuda()
    b = torch.randn(N).cuda(); g = torch.randn(N).cuda(); be = torch.randn(N).cuda(); hw = torch.randn(N).cuda()
    outs, preds = [], []
    for nb in (32, 8, 1):
        if ln:
            y, pr = _gpu.gemm_ln(dt, x[:nb * S].contiguous(), w, b, None, g, be, taps=taps, S=S, relu=True, dot_w=hw, dot_b=0.3)
            preds.append(pr[:S])
        else:
            y = _gpu.gemm(dt, x[:nb * S].contiguous(), w, b, taps=taps, S=S, relu=True)
        outs.append(y[:S].float())
    print(name, [bool(torch.equal(outs[0], o)) for o in outs], [bool(torch.equal(preds[0], q)) for q in preds])
