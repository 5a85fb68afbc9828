#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into a per-kernel table:
calls, total ms, average us, min/max us, share of GPU kernel time.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels "
                       f"group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % of kernel time |", "|---|---|---|---|---|---|---|"]
    for n, c, t, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append(f"| `{short}` | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / total:.2f} |")
    lines.append(f"\ntotal kernel time: {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
