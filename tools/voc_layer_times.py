#!/usr/bin/env python
"""Per-layer times of one HiFi-GAN pass from a rocprofv3 rocpd database (kernel trace of
tools/bench_vocoder.py): the last 78 vocoder_conv dispatches, in launch order, labelled by layer."""
import sqlite3
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningfastspeech2_amd.hifigan import HifiGanConfig


def fused_mi16(C, k, dils, esz=2):
    """Mirror of voc_resblock_mi16 (vocoder_resblock.hip): tile height or 0."""
    if C not in (32, 64, 128):
        return 0
    if len(dils) == 3 and k * C > 224:
        return 0
    c = (k - 1) // 2
    H, G = c * (sum(d + 1 for d in dils) - dils[0]), c * max(dils)
    for mi in (8, 4):
        R = (8 // (C // 32)) * mi * 16
        if ((R + 2 * G) + (R + 2 * c)) * C * esz <= 150 * 1024 and (R - 2 * H) * 5 >= R * 4:
            return mi
    return 0


def labels(cfg, fused=True, blocks=True):
    """One (label, flops per frame) per launch of a pass, in launch order (vocoder_engine.hip)."""
    ch = cfg.channels()
    out = [("conv_pre", 2.0 * 80 * ch[0] * 7, 1)]
    up = 1
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        out.append((f"ups{i} {ch[i]}->{ch[i+1]} x{u}", up * u * 2.0 * ch[i] * ch[i + 1] * (k / u), up))
        up *= u
        C = ch[i + 1]
        for rk, rd in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes):
            conv = up * 2.0 * C * C * rk
            if fused and blocks and fused_mi16(C, rk, rd):
                out.append((f"s{i} C={C} k={rk} block", 6 * conv, up))
                continue
            for d in rd:
                if fused and fused_mi16(C, rk, [d]):
                    out.append((f"s{i} C={C} k={rk} pair d={d}", 2 * conv, up))
                else:
                    out.append((f"s{i} C={C} k={rk} d={d}", conv, up))
                    out.append((f"s{i} C={C} k={rk} d=1 +res", conv, up))
    out.append(("conv_post", up * 2.0 * ch[-1] * 7, up))
    return out


def main(db, frames, fused=True, blocks=True):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels where name like '%vocoder_%' order by start").fetchall()
    lab = labels(HifiGanConfig(), fused, blocks)
    rows = rows[-len(lab):]
    tot = 0.0
    agg = {}
    for (name, s, e), (l, fl, up) in zip(rows, lab):
        us = (e - s) / 1e3
        tot += us
        key = l.split(" d=")[0].replace(" block", "").replace(" pair", "") if l.startswith("s") else l
        a = agg.setdefault(key, [0.0, 0.0])
        a[0] += us
        a[1] += fl * frames
    print(f"one pass: {tot/1e3:.2f} ms kernel time over {len(rows)} launches")
    for k, (us, fl) in agg.items():
        print(f"{k:28s} {us:9.1f} us  {fl/us/1e6:8.1f} TF  ({fl/us/1e6/2500*100:4.1f}% of 2.5 PF)  {100*us/tot:5.1f}% of pass")


if __name__ == "__main__":
    mode = sys.argv[3] if len(sys.argv) > 3 else "fused"  # fused | pairs (no whole-block launches) | unfused
    main(sys.argv[1], float(sys.argv[2]), fused=mode != "unfused", blocks=mode == "fused")
