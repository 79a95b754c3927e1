#!/usr/bin/env python
"""Per-queue timeline statistics of a rocprofv3 kernel trace (rocpd sqlite): busy time, gaps between consecutive kernels of one
queue (the dependent-launch latency a hipGraph chain pays), concurrency.   python tools/rocpd_gaps.py x_results.db"""
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    rows = cur.execute("select queue_id, stream_id, start, end from kernels order by start").fetchall()
    perq = defaultdict(list)
    for q, s, a, b in rows:
        perq[(q, s)].append((a, b))
    t0, t1 = min(r[2] for r in rows), max(r[3] for r in rows)
    print("| queue, stream | kernels | busy ms | gaps<100us: count | mean gap us | median gap us | sum gap ms | span ms |")
    print("|---|---|---|---|---|---|---|---|")
    for key, ks in sorted(perq.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        busy = sum(b - a for a, b in ks) / 1e6
        gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
        small = sorted(g for g in gaps if 0 <= g < 100)
        if not small:
            continue
        print("| %s | %d | %.2f | %d | %.2f | %.2f | %.2f | %.2f |" % (key, len(ks), busy, len(small), sum(small) / len(small), small[len(small) // 2],
                                                                 sum(small) / 1e3, (ks[-1][1] - ks[0][0]) / 1e6))
    # concurrency histogram over the whole span (sweep line)
    ev = []
    for q, s, a, b in rows:
        ev.append((a, 1)); ev.append((b, -1))
    ev.sort()
    lvl, last, hist = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        hist[lvl] += t - last
        last = t
        lvl += d
    tot = sum(hist.values())
    print("\nconcurrency (kernels running at once) as share of the traced span %.1f ms:" % ((t1 - t0) / 1e6), {k: round(v / tot, 3) for k, v in sorted(hist.items())})


if __name__ == "__main__":
    main()
