#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints: calls, total/avg/min/max duration, share.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--by-grid] > profiles/x.md
--by-grid: one row per (kernel, grid size) - separates the shapes a GEMM kernel is launched with."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if c in cols]
    wcols = [c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
    sel = name_col + ", start, end" + ("".join(", " + c for c in gcols + wcols[:1]) if by_grid else "")
    rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
    if by_grid:
        print("<!-- kernels columns: %s -->" % cols)
    agg = {}
    for r in rows:
        name, s, e = r[0], r[1], r[2]
        key = name
        if by_grid and len(r) > 3:
            g = [int(x or 1) for x in r[3:3 + len(gcols)]]
            wg = int(r[3 + len(gcols)] or 1) if len(r) > 3 + len(gcols) else 1
            threads = 1
            for x in g:
                threads *= x
            key = "%s [grid %s, %d wg]" % (name if len(name) < 90 else name[:87] + "...", "x".join(str(x) for x in g), threads // max(wg, 1))
        d = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 150 else name[:147] + "..."
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (short, v[0], v[1], v[1] / v[0], v[2], v[3], 100 * v[1] / tot))
    print("\ntotal kernel time: %.1f us over %d dispatches" % (tot, len(rows)))


if __name__ == "__main__":
    main()
