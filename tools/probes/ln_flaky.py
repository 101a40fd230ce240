"""Repeat the fused GEMM+LN op on one shape and report run-to-run differences (race hunting)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpu as G

def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale

B, S, Cin, N, k = 2, 333, 1024, 256, 1
x = rnd(B, S, Cin, seed=50); w = rnd(N, Cin, k, seed=51, scale=(Cin * k) ** -0.5)
b, res = rnd(N, seed=52), rnd(B * S, N, seed=53)
g, be = 1 + 0.2 * rnd(N, seed=54), 0.1 * rnd(N, seed=55)
for dtype in (G.F32, G.BF16):
    z = F.conv1d(G.rounded(x, dtype).transpose(1, 2), G.rounded(w, dtype), b, padding="same").transpose(1, 2).reshape(B * S, N)
    ref = F.layer_norm(z + G.rounded(res, dtype), (N,), g, be, 1e-5)
    for variant in (6, 7, 3, 4):
        G.lib().fs2_op_set_gemm_variant(variant)
        bad = []
        for it in range(40):
            y, _ = G.gemm_ln(dtype, x.reshape(B * S, Cin), G.pack_conv_weight(w), b, res, g, be, taps=k, S=S, relu=False)
            d = (y - ref).abs()
            if float(d.max()) > 0.05:
                rows = torch.nonzero(d.max(dim=1).values > 0.05).flatten().tolist()
                bad.append((it, round(float(d.max()), 3), rows[:6], len(rows)))
        print("dtype", dtype, "variant", variant, "bad runs", len(bad), bad[:5], flush=True)
G.lib().fs2_op_set_gemm_variant(0)
