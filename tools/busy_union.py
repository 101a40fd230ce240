#!/usr/bin/env python
"""GPU-busy fraction of a rocprofv3 kernel trace: union of the kernel intervals (all streams) over wall time, and the mean number of
kernels resident while the GPU is busy (sum of durations / union).  Window = the forwards between two phone-embedding launches
(embed_kernel opens a forward): [--first I, --last J] by index into the trace's embed launches (default: the middle half).
    python tools/busy_union.py results.db [--first I --last J]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--first", type=int, default=-1)
    ap.add_argument("--last", type=int, default=-1)
    ap.add_argument("--title", default="")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    emb = [r[1] for r in rows if "embed_kernel<" in r[0] and "bucket" not in r[0]]
    i0 = a.first if a.first >= 0 else len(emb) // 4
    i1 = a.last if a.last >= 0 else (3 * len(emb)) // 4
    w0, w1 = emb[i0], emb[i1]
    iv = sorted((max(s, w0), min(e, w1)) for _, s, e in rows if e > w0 and s < w1)
    union, busy_sum, cur_s, cur_e = 0, 0, None, None
    for s, e in iv:
        busy_sum += e - s
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    wall = w1 - w0
    nf = i1 - i0
    print(f"{a.title or a.db}: {nf} forwards in {wall / 1e6:.3f} ms = {wall / nf / 1e6:.3f} ms per forward; GPU busy (union of kernel intervals) "
          f"{100 * union / wall:.1f} % of the wall time, idle {(wall - union) / nf / 1e3:.1f} us per forward; sum of kernel durations "
          f"{busy_sum / nf / 1e6:.3f} ms per forward = {busy_sum / union:.2f} kernels resident on average while busy")


if __name__ == "__main__":
    main()
