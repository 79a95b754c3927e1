#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints: calls, total/avg/min/max duration, share.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-first-frac 0.3] > profiles/x.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (short, v[0], v[1], v[1] / v[0], v[2], v[3], 100 * v[1] / tot))
    print("\ntotal kernel time: %.1f us over %d dispatches" % (tot, len(rows)))


if __name__ == "__main__":
    main()
