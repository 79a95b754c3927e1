#!/usr/bin/env python
"""One optimisation step of train.py:186-286 on a KITTI-sized synthetic frame (20 480 points, 160 x 512 image, num_kpt 64), timed on
the GPU box (tool, not a test):  python tools/train_bench.py [--steps 10] [--warmup 3] [--points 20480] [--arith bf16x6] [--stress]
Prints one JSON line: ms per step (forward / backward / optimizer split from device events), peak memory, the loss trajectory."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=20480)
    ap.add_argument("--arith", default="bf16x6")
    ap.add_argument("--graph-only", action="store_true", help="skip the eager steps (a clean kernel trace of the recorded step)")
    ap.add_argument("--stress", action="store_true", help="BASELINE configs[4] shape: 896 x 1600 image, 40 960 points")
    args = ap.parse_args()
    if args.stress:
        bench.Opt.img_H, bench.Opt.img_W, args.points = 896, 1600, 40960
    dev = torch.device("cuda", 0)
    frame, = bench.make_inputs(dev, [0], args.points)
    out = bench.train_step_summary(dev, frame, steps=args.steps, warmup=args.warmup, arith=args.arith, eager=not args.graph_only)
    out.update(metric="train_step_ms", points=args.points)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
