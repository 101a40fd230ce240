#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`, which
on ROCm 7.2 writes NAME_results.db) into a per-kernel markdown table for profiles/."""
import re
import sqlite3
import sys


def main(db, title=""):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {title or db}\n")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx, vg, ag, lds in rows:
        n = re.sub(r"fs2::", "", n)[:100]
        print(f"| `{n}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.1f} | {vg} | {ag} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
