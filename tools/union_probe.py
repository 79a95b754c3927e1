#!/usr/bin/env python
"""GPU tool: how much the 128-neighbour rows of spatially adjacent queries overlap on the bench frames - the distinct source rows of groups of
G consecutive queries (Morton order, as the gather kernels process them) per pyramid stage, for the neighbour tables (KPConv aggregation)
and the sub-sampling tables (max-pool).  A gather kernel that loads the UNION of a group once reads union / (128 G) of today's rows.
    python tools/union_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from cofii2p_amd.preprocess import morton_order


def union_stats(table, order, G, sample=4096):
    t = table[order.long()][: (table.shape[0] // G) * G].reshape(-1, G * table.shape[1])
    if t.shape[0] > sample:
        t = t[torch.randperm(t.shape[0], device=t.device)[:sample]]
    s = torch.sort(t, dim=1)[0]
    distinct = 1 + (s[:, 1:] != s[:, :-1]).sum(1)
    return float(distinct.float().mean()), int(distinct.max())


def main():
    dev = torch.device("cuda", 0)
    pyr = bench.make_inputs(dev, [0], 20480)[0][0]
    pts = pyr["points"]
    for name, tabs, qstage in (("neighbors (KPConv aggregation)", pyr["neighbors"], 0), ("subsampling (max-pool)", pyr["subsampling"], 1)):
        for i, tab in enumerate(tabs):
            order = morton_order(pts[i + qstage])
            cells = []
            for G in (2, 4, 8, 16):
                mean, mx = union_stats(tab, order, G)
                cells.append("G=%-2d %6.1f rows (%.2f of %d, max %d)" % (G, mean, mean / (G * tab.shape[1]), G * tab.shape[1], mx))
            print("%-32s stage %d  %6d queries | %s" % (name, i, tab.shape[0], "  ".join(cells)))


if __name__ == "__main__":
    main()
