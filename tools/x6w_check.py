#!/usr/bin/env python
"""GPU tool: bf16x6 with pre-split three-plane weights (ops.X6_W_SPLIT) against the on-the-fly split - bit identity + per-launch time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cofii2p_amd import ops
from tools.gemm_shapes import time_graph

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
ops.GEMM_MODE = "bf16x6"
bad = 0
for M, N, K in [(1280, 128, 128), (1280, 512, 7680), (2560, 1024, 3072), (20480, 64, 576), (320, 256, 2304), (77, 33, 60), (20480, 128, 32), (20480 * 4, 512, 1536),
                (327680, 128, 480), (40960, 1024, 3072)]:
    a, w, bias = rn(M, K), ops.presplit(rn(N, K) / K ** 0.5), rn(N)
    res = {}
    for mode in (False, True):
        ops.X6_W_SPLIT = mode
        y, part = ops.gemm_colstats(a, w, bias=bias, act=ops.ACT_LEAKY01)
        t = time_graph(lambda: ops.gemm_colstats(a, w, bias=bias, act=ops.ACT_LEAKY01), reps=10)
        res[mode] = (y, part, t)
    same = torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    bad += not same
    print("gemm %7d %5d %5d  %8.1f -> %8.1f us  %s" % (M, N, K, 1e6 * res[False][2], 1e6 * res[True][2], "bit-equal" if same else "MISMATCH"), flush=True)
for H, W, Cin, Cout, ks, st in [(40, 128, 64, 64, 3, 1), (20, 64, 128, 128, 3, 1), (160, 512, 64, 128, 3, 1), (20, 64, 128, 256, 1, 2)]:
    x, wt = rn(H * W, Cin), ops.presplit(rn(Cout, ks * ks * Cin) / (ks * ks * Cin) ** 0.5)
    y0, part0 = ops.gemm_colstats(rn(H * W, 64), rn(Cin, 64) / 8)
    nx = ops.Normed(y0, ops.ColStats(part0, H * W, Cin), slope=0.0)
    res = {}
    for mode in (False, True):
        ops.X6_W_SPLIT = mode
        ya = ops.conv2d_nhwc(x, H, W, wt, ks, st, ks // 2, colstats=True)
        yb = ops.conv2d_nhwc(nx, H, W, wt, ks, st, ks // 2, colstats=True)
        t = time_graph(lambda: ops.conv2d_nhwc(x, H, W, wt, ks, st, ks // 2, colstats=True), reps=10)
        res[mode] = (ya[0], ya[1], yb[0], yb[1], t)
    same = all(torch.equal(res[False][i], res[True][i]) for i in range(4))
    bad += not same
    print("conv %dx%d %d->%d k%d s%d  %8.1f -> %8.1f us  %s" % (H, W, Cin, Cout, ks, st, 1e6 * res[False][4], 1e6 * res[True][4], "bit-equal" if same else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
