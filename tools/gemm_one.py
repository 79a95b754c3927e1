#!/usr/bin/env python
"""GPU tool: ONE dense bf16x6 contraction, repeated (profiling target for rocprofv3 passes, and a timer).
    python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --reps 5 [--time]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="40960x1024x3072")
    ap.add_argument("--kernel", default="big", choices=["big", "small", "auto"])
    ap.add_argument("--ks", type=int, default=0)
    ap.add_argument("--bm", type=int, default=0, help="small: forced tile rows (0 = planner)")
    ap.add_argument("--bn", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--colstats", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--presplit", action="store_true", help="W as a static operand (ops.SplitW): the f16x3 kernel reads it pre-split (COFI_GEMM_W_F16PRE)")
    ap.add_argument("--dbg", type=int, default=0, help="cofi_tune_big_debug flags (timing experiments: results are wrong)")
    args = ap.parse_args()
    from cofii2p_amd import _lib, ops

    ops.GEMM_MODE = "bf16x6"
    lib = _lib.load()
    fp, fb = lib.cofi_tune_force_plan, lib.cofi_tune_force_big
    fp.argtypes, fp.restype = [ctypes.c_int] * 3, ctypes.c_int
    fb.argtypes, fb.restype = [ctypes.c_int] * 2, ctypes.c_int
    M, N, K = (int(x) for x in args.shape.split("x"))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, generator=g, device=dev)
    w = torch.randn(N, K, generator=g, device=dev) / K ** 0.5
    lib.cofi_tune_big_debug.argtypes, lib.cofi_tune_big_debug.restype = [ctypes.c_int], ctypes.c_int
    lib.cofi_tune_big_debug(args.dbg)
    if args.kernel == "big":
        fb(1, args.ks)
    elif args.kernel == "small":
        fb(-1, 0)
        if args.bm:
            fp(args.bm, args.bn, args.ks)
    if args.presplit:
        w = ops.presplit(w)
    fn = (lambda: ops.gemm_colstats(a, w)) if args.colstats else (lambda: ops.gemm(a, w))
    fn()
    torch.cuda.synchronize()
    if args.time:
        from tools.gemm_shapes import time_graph

        t = time_graph(fn, reps=args.reps)
        print("%s %s ks=%d dbg=%d: %.1f us  %.1f TF/s" % (args.shape, args.kernel, args.ks, args.dbg, t * 1e6, 2.0 * M * N * K / t * 1e-12))
    else:
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        print("ran %d" % args.reps)


if __name__ == "__main__":
    main()
