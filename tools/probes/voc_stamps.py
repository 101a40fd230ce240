#!/usr/bin/env python
"""Phase timeline of the vocoder kernels from s_memtime stamps (probe build only):
    tools/build_variant_files.sh stamp <stamped vocoder_conv.hip> <stamped vocoder_resblock.hip>
    FS2_LIB=.../variants/libfs2_stamp.so python tools/probes/voc_stamps.py
Wave 0 of eight workgroups spread over each launch stamps: start, slab filled (after the barrier), end
of every K loop, end of every epilogue (+ barrier).  Printed: mean shader cycles per phase and the
clock implied by the 100 MHz wall counter."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from lightningfastspeech2_amd import _lib
from lightningfastspeech2_amd.hifigan import HifiGan, HifiGanConfig, synth_state_dict

cfg = HifiGanConfig()
g = HifiGan(cfg, synth_state_dict(cfg, 0), precision="bf16")
if len(sys.argv) > 1:
    _lib.load().fs2_op_set_vocoder_fused_resblock(int(sys.argv[1]))
rs = np.random.RandomState(1234)
mel = torch.from_numpy((rs.standard_normal((32, 1536, 80)) * 1.5 - 4.0).astype(np.float32)).cuda()
for _ in range(3):
    g.synthesize(mel)
torch.cuda.synchronize()
lib = _lib.load()

rb = np.zeros(48 * 8 * 16, dtype=np.uint64)
assert lib.fs2_dbg_rb_stamps(ctypes.c_void_p(rb.ctypes.data)) == 0
rb = rb.reshape(48, 8, 16).astype(np.int64)
print("resident resblock kernel: cycles per phase (mean of 8 workgroups)")
for ci in range(36):
    s = rb[ci]
    if not s[:, 0].any():
        continue
    C = (32, 64, 128)[ci // 12]
    k = 3 + 4 * ((ci // 4) % 3)
    mode = "block" if ci % 4 == 3 else f"pair d={1 + 2 * (ci % 4)}"
    n = 14 if ci % 4 == 3 else 6
    d = np.diff(s[:, :n], axis=1).mean(axis=0)
    tot = (s[:, n - 1] - s[:, 0]).mean()
    wall = (s[:, 14] - s[:, 15]).mean()  # 100 MHz ticks
    ghz = tot / (wall * 10.0) if wall > 0 else 0
    lab = ["fill"] + [x for j in range((n - 2) // 2) for x in (f"K{j}", f"E{j}")]
    print(f"C={C:3d} k={k:2d} {mode:9s} total {tot:8.0f} cyc ({wall/100:6.1f} us, {ghz:4.2f} GHz): " +
          " ".join(f"{l}={v:.0f}" for l, v in zip(lab, d)))
    if n == 6:  # pairs: when each of the 8 waves left K loop 0, relative to "slab filled"
        print("        waves leave K0 at: " + " ".join(f"{v:.0f}" for v in (s[:, 6:14] - s[:, 1:2]).mean(axis=0)))

cv = np.zeros(64 * 8 * 8, dtype=np.uint64)
assert lib.fs2_dbg_cv_stamps(ctypes.c_void_p(cv.ctypes.data)) == 0
cv = cv.reshape(64, 8, 8)
print("conv kernel: cycles per phase")
for ci in range(64):
    s = cv[ci]
    if not s[:, 0].any():
        continue
    key = int(s[s[:, 0] != 0][0, 6])
    cin, n, taps, dil, mi, flags = key >> 48, (key >> 32) & 0xffff, (key >> 24) & 0xff, (key >> 16) & 0xff, (key >> 8) & 0xff, key & 0xff
    s = s[s[:, 0] != 0].astype(np.int64)
    d = np.diff(s[:, :4], axis=1).mean(axis=0)
    tot = (s[:, 3] - s[:, 0]).mean()
    wall = (s[:, 4] - s[:, 5]).mean()
    ghz = tot / (wall * 10.0) if wall > 0 else 0
    print(f"cin={cin:3d} n={n:4d} k={taps:2d} d={dil} mi16={mi:2d} res={flags>>1} acc={flags&1} total {tot:8.0f} cyc ({wall/100:6.1f} us, {ghz:4.2f} GHz): "
          f"fill={d[0]:.0f} K={d[1]:.0f} epi={d[2]:.0f}")
