#!/usr/bin/env python
"""Per-kernel PMC counter averages from a rocprofv3 (rocpd sqlite) run with --pmc.
    python tools/rocpd_pmc_summary.py x_results.db > summary.md"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if not cols:
        print("no counters_collection view; tables:", [r[0] for r in cur.execute("select name from sqlite_master")])
        return
    print("<!-- counters_collection columns: %s -->" % cols)
    kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else [c for c in cols if "kernel" in c][0])
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    rows = cur.execute("select %s, %s, count(*), sum(%s), avg(%s) from counters_collection group by %s, %s" % (kcol, ccol, vcol, vcol, kcol, ccol)).fetchall()
    print("| kernel | counter | dispatches | sum | avg per dispatch |")
    print("|---|---|---|---|---|")
    for k, c, n, s, a in sorted(rows, key=lambda r: (-(r[3] or 0))):
        k = k if len(k) < 100 else k[:97] + "..."
        print("| `%s` | %s | %d | %.6g | %.6g |" % (k, c, n, s or 0, a or 0))


if __name__ == "__main__":
    main()
