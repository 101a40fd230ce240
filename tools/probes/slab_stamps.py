#!/usr/bin/env python
"""When does each of the 8 waves of a decoder-conv1 workgroup reach / leave the per-step barrier?  (probe build:
gemm_mfma.hip with s_memtime stamps around dma_drain + __syncthreads, -> variants/libfs2_slabstamp.so)
    FS2_LIB=.../variants/libfs2_slabstamp.so python tools/probes/slab_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _gpu

torch.manual_seed(0)
S, K, N, taps = 1536, 256, 1024, 9
x = torch.randn(32 * S, K); w = torch.randn(N, K * taps) / (K * taps) ** 0.5; b = torch.randn(N)
for _ in range(3):
    _gpu.gemm(_gpu.BF16, x, w, b, taps=taps, S=S, relu=True)
st = np.zeros(8 * 128 * 3, dtype=np.uint64)
assert _gpu.lib().fs2_dbg_slab_stamps(ctypes.c_void_p(st.ctypes.data)) == 0
st = st.reshape(8, 128, 3).astype(np.int64)
nsteps = 36
t0 = st[:, 0, 0].min()
arr, drained, rel = st[:, :nsteps, 0] - t0, st[:, :nsteps, 1] - t0, st[:, :nsteps, 2] - t0
print("step: release(mean)  | arrival at barrier minus the step's release, per wave 0..7 (cycles) | drain wait")
for s_ in range(2, 14):
    r_prev = rel[:, s_ - 1].mean()
    print(f"{s_:3d}: step length {rel[:, s_].mean() - r_prev:6.0f} | " + " ".join(f"{arr[w_, s_] - r_prev:6.0f}" for w_ in range(8)) +
          " | " + " ".join(f"{drained[w_, s_] - arr[w_, s_]:5.0f}" for w_ in range(8)))
steps = np.diff(rel.mean(axis=0))
work = (arr[:, 1:nsteps] - rel[:, :nsteps - 1])
print(f"mean step {steps[2:].mean():.0f} cycles; wave work time (release -> next arrival): older waves 0-3 {work[:4, 2:].mean():.0f}, younger 4-7 {work[4:, 2:].mean():.0f};"
      f" drain wait {(drained - arr)[:, 2:nsteps].mean():.0f}; barrier wait older {(rel - drained)[:4, 2:nsteps].mean():.0f} younger {(rel - drained)[4:, 2:nsteps].mean():.0f}")
print(f"ideal MFMA time per step per SIMD: 2 waves x 64 MFMAs x 16 = 2048 cycles")
