#!/usr/bin/env python
"""One gemm_planes_kernel configuration on one shape, R launches (GPU tool for rocprofv3 passes and ablation timings).
    python tools/planes_probe.py --shape 20480,512,7680 --cfg 0 --ks 2 [--reps 5] [--time]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cofii2p_amd import _lib, ops
from tools.planes_bench import split_a, time_graph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="20480,512,7680")
    ap.add_argument("--cfg", default="0")
    ap.add_argument("--ks", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    ops.GEMM_MODE = "bf16x3"
    lib = _lib.load()
    lib.cofi_tune_force_planes.argtypes, lib.cofi_tune_force_planes.restype = [ctypes.c_int] * 2, ctypes.c_int
    dev = torch.device("cuda", 0)
    M, N, K = (int(x) for x in args.shape.split(","))
    a = torch.randn((M, K), device=dev)
    w = ops.SplitW(torch.randn((N, K), device=dev) / K ** 0.5)
    sa = split_a(a)
    del a
    for cfg in [int(c) for c in args.cfg.split(",")]:
        lib.cofi_tune_force_planes(cfg, args.ks)
        run = lambda: ops.gemm(sa, w)
        if args.time:
            t = time_graph(run, reps=10)
            print("shape %s cfg %d ks %d: %.1f us  (%.0f TF/s)" % (args.shape, cfg, args.ks, 1e6 * t, 2.0 * M * N * K / t / 1e12), flush=True)
        else:
            for _ in range(args.reps):
                run()
            torch.cuda.synchronize()
    lib.cofi_tune_force_planes(-1, 0)


if __name__ == "__main__":
    main()
