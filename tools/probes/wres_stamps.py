#!/usr/bin/env python
"""Per-tile phases of workgroup 0 of the weight-resident GEMM (probe build -DFS2_WRES_PROBE of gemm_wres.hip linked into
variants/libfs2_wresprobe.so): wait for the tile's DMA | barrier | DMA + store issue | K loop | epilogue."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FS2_LIB"] = os.path.join(ROOT, "lightningfastspeech2_amd", "variants", "libfs2_wresprobe.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _gpu
torch.manual_seed(0)
M, N = 49152, 768
x = torch.randn(M, 256); w = torch.randn(N, 256) / 16; b = torch.randn(N)
for _ in range(3):
    _gpu.gemm(_gpu.BF16, x, w, b)
st = np.zeros(128, dtype=np.uint64)
assert _gpu.lib().fs2_dbg_wres_stamps(ctypes.c_void_p(st.ctypes.data)) == 0
st = st.reshape(2, 64).astype(np.int64)
for wv in range(2):
    s = st[wv]
    print(f"wave {0 if wv == 0 else 7}: index math {s[1] - s[0]} (entry -> first DMA incl. weight staging)")
    for i in range(7):
        b_ = 4 + 5 * i
        if s[b_ + 4] == 0: break
        nxt = s[b_ + 5] if s[b_ + 5] else s[b_ + 4]
        print(f"  tile {i}: drain {s[b_ + 1] - s[b_]:6d} | barrier {s[b_ + 2] - s[b_ + 1]:6d} | issue {s[b_ + 3] - s[b_ + 2]:6d} | K loop {s[b_ + 4] - s[b_ + 3]:6d} | epilogue {nxt - s[b_ + 4]:6d}")
