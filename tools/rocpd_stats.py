#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`, which
on ROCm 7.2 writes NAME_results.db) into a per-kernel markdown table for profiles/."""
import re
import sqlite3
import sys


def by_grid(db, title=""):
    """One row per (kernel, grid) - the launch classes of a forward, for configs whose step is many shapes of one kernel."""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    g = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if c in cols][:3]
    rows = con.execute(f"select name, {', '.join(g)}, count(*), sum(end-start), avg(end-start), min(end-start) from kernels "
                       f"group by name, {', '.join(g)} order by {len(g) + 3} desc").fetchall()
    tot = sum(r[len(g) + 2] for r in rows)
    print(f"# {title or db} (by launch class)\n")
    print("| kernel | grid | calls | total ms | avg us | min us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows[:40]:
        n = re.sub(r"fs2::", "", r[0])[:90]
        c, s_, a, mn = r[len(g) + 1:]
        print(f"| `{n}` | {'x'.join(str(x) for x in r[1:len(g) + 1])} | {c} | {s_ / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {100 * s_ / tot:.1f} |")


def main(db, title=""):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {title or db}\n")
    nfw = sum(r[1] for r in rows if "embed_kernel<" in r[0] and "bucket" not in r[0])  # the phone embedding opens a forward
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f"; {nfw} forwards in the trace (warm-up, launch-mode tuning, the timed steps and bench.py's event-bracketed passes)\n" if nfw else "\n"))
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx, vg, ag, lds in rows:
        n = re.sub(r"fs2::", "", n)[:100]
        print(f"| `{n}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.1f} | {vg} | {ag} | {lds} |")

def timeline(db, title=""):
    """The launch sequence of ONE forward (the last complete one in the trace): per launch its duration and the idle gap since the
    previous launch ended - which launch is which GEMM, and where the GPU waits for the host."""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    g = [c for c in ("grid_x", "grid_size_x") if c in cols][0]
    rows = con.execute(f"select name, start, end, {g} from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "embed_kernel<" in r[0] and "bucket" not in r[0]]  # the phone embedding opens a forward
    # one forward at a time = the launch count between two phone embeddings that occurs most often (the bench's pipelined sections
    # interleave two forwards, its tail sections run other things): the last such pair
    from collections import Counter
    diffs = [idx[i + 1] - idx[i] for i in range(len(idx) - 1)]
    common = Counter(diffs).most_common(1)[0][0]
    j = max(i for i, d in enumerate(diffs) if d == common)
    a, b = idx[j], idx[j + 1]
    print(f"# {title or db}: launches of one forward ({b - a} launches, {(rows[b][1] - rows[a][1]) / 1e3:.1f} us start to start)\n")
    print("| # | kernel | grid x | us | gap before, us |")
    print("|---|---|---|---|---|")
    busy = 0
    for i in range(a, b):
        n = re.sub(r"fs2::|\(anonymous namespace\)::|void ", "", rows[i][0])[:70]
        busy += rows[i][2] - rows[i][1]
        print(f"| {i - a} | `{n}` | {rows[i][3]} | {(rows[i][2] - rows[i][1]) / 1e3:.1f} | {(rows[i][1] - rows[i - 1][2]) / 1e3:.1f} |")
    print(f"\nkernel time {busy / 1e3:.1f} us, idle {(rows[b][1] - rows[a][1] - busy) / 1e3:.1f} us")



if __name__ == "__main__":
    if sys.argv[1] == "--by-grid":
        by_grid(sys.argv[2], " ".join(sys.argv[3:]))
    elif sys.argv[1] == "--timeline":
        timeline(sys.argv[2], " ".join(sys.argv[3:]))
    else:
        main(sys.argv[1], " ".join(sys.argv[2:]))
