#!/usr/bin/env python
"""Where does a fused GEMM + LayerNorm launch spend its time?  Probe build of gemm_mfma.hip (-DFS2_SLAB_PROBE): s_memtime of
waves 0 and 7 of workgroups 0 and 128 at kernel entry, before the first operand DMA, after the K loop, after the barrier that
follows it, before the stores and after them.
    EXTRA=-DFS2_SLAB_PROBE tools/build_variant.sh slabprobe   (here, then)   gpurun -- python tools/probes/slab_phase_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FS2_LIB"] = os.path.join(ROOT, "lightningfastspeech2_amd", "variants", "libfs2_slabprobe.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _gpu

torch.manual_seed(0)
for name, M, K, N, S in [("decoder out-proj + res + LN", 49152, 256, 256, 1536), ("decoder conv2 + res + LN", 49152, 1024, 256, 1536),
                         ("encoder out-proj + res + LN", 8192, 256, 256, 256), ("encoder conv2 + res + LN", 8192, 1024, 256, 256)]:
    x = torch.randn(M, K); w = torch.randn(N, K) / K ** 0.5; b = torch.randn(N); res = torch.randn(M, N)
    for _ in range(3):
        _gpu.gemm_ln(_gpu.BF16, x, w, b, res, torch.ones(N), torch.zeros(N), S=S)
    st = np.zeros(32, dtype=np.uint64)
    assert _gpu.lib().fs2_dbg_slab_phase_stamps(ctypes.c_void_p(st.ctypes.data)) == 0
    st = st.reshape(4, 8).astype(np.int64)
    print(name)
    for i, who in enumerate(["wg 0 wave 0", "wg 0 wave 7", "wg 128 wave 0", "wg 128 wave 7"]):
        d = np.diff(st[i, :6])
        d[1] = st[i, 2] - st[i, 6]
        print(f"  {who}: prologue {d[0]:6d} | residual preload {st[i, 6] - st[i, 1]:6d} | K loop {d[1]:6d} | barrier {d[2]:5d} | statistics {d[3]:6d} | normalise + store issue {d[4]:6d}   ticks;"
              f"  entry after wg 0's {st[i, 0] - st[0, 0]:6d}")
