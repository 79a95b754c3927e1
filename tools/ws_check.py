#!/usr/bin/env python
"""GPU tool: the warp-specialised bf16-split GEMM kernel (COFI_GEMM_WS) against the single-role kernel - bit identity on dense, normalising-loader
and convolution launches - and per-launch times of both."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cofii2p_amd import _lib, ops
from tools.gemm_shapes import time_graph

lib = _lib.load()
force = lib.cofi_tune_force_ws
force.argtypes, force.restype = [ctypes.c_int], ctypes.c_int
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
bad = 0
for mode, bit in (("bf16x6", 1), ("bf16x3", 2)):
    ops.GEMM_MODE = mode
    print("==", mode)
    for M, N, K in [(1280, 128, 128), (1280, 512, 7680), (2560, 1024, 3072), (20480, 64, 576), (320, 256, 2304), (5120, 512, 1536), (77, 33, 60), (20480, 128, 32),
                    (1280, 2048, 512), (10240, 64, 960), (1280, 1, 64)]:
        a, w, bias = rn(M, K), ops.presplit(rn(N, K) / K ** 0.5), rn(N)
        rd = torch.randint(1, 9, (M,), device=dev).float()
        res = {}
        for ws in (0, bit):
            force(ws)
            y, part = ops.gemm_colstats(a, w, bias=bias, rowdiv=rd, act=ops.ACT_LEAKY01)
            t = time_graph(lambda: ops.gemm_colstats(a, w, bias=bias, rowdiv=rd, act=ops.ACT_LEAKY01), reps=10)
            res[ws] = (y, part, t)
        same = torch.equal(res[0][0], res[bit][0]) and torch.equal(res[0][1], res[bit][1])
        bad += not same
        print("gemm %6d %5d %5d  %7.1f -> %7.1f us  %s" % (M, N, K, 1e6 * res[0][2], 1e6 * res[bit][2], "bit-equal" if same else "MISMATCH"), flush=True)
    # normalising loader + convolutions
    for H, W, Cin, Cout, ks, st in [(40, 128, 64, 64, 3, 1), (20, 64, 128, 128, 3, 1), (40, 128, 64, 128, 3, 2), (10, 32, 256, 256, 3, 1), (80, 256, 64, 64, 3, 1), (20, 64, 128, 256, 1, 2)]:
        x, wt = rn(H * W, Cin), ops.presplit(rn(Cout, ks * ks * Cin) / (ks * ks * Cin) ** 0.5)
        y0, part0 = ops.gemm_colstats(rn(H * W, 64), rn(Cin, 64) / 8)   # a producer with statistics: its output feeds the normalising loader
        nx = ops.Normed(y0, ops.ColStats(part0, H * W, Cin), slope=0.0)
        res = {}
        for ws in (0, bit):
            force(ws)
            ya = ops.conv2d_nhwc(x, H, W, wt, ks, st, ks // 2, colstats=True)
            yb = ops.conv2d_nhwc(nx, H, W, wt, ks, st, ks // 2, colstats=True)
            t = time_graph(lambda: ops.conv2d_nhwc(nx, H, W, wt, ks, st, ks // 2, colstats=True), reps=10)
            res[ws] = (ya[0], ya[1], yb[0], yb[1], t)
        same = all(torch.equal(res[0][i], res[bit][i]) for i in range(4))
        bad += not same
        print("conv %dx%d %d->%d k%d s%d  %7.1f -> %7.1f us  %s" % (H, W, Cin, Cout, ks, st, 1e6 * res[0][4], 1e6 * res[bit][4], "bit-equal" if same else "MISMATCH"), flush=True)
force(-1)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
