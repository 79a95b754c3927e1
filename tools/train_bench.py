#!/usr/bin/env python
"""One optimisation step of train.py:186-286 on a KITTI-sized synthetic frame (20 480 points, 160 x 512 image, num_kpt 64), timed on
the GPU box (tool, not a test):  python tools/train_bench.py [--steps 10] [--warmup 3] [--points 20480] [--arith f32]
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
    ap.add_argument("--arith", default="f32")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    frame, = bench.make_inputs(dev, [0], args.points)
    out = bench.train_step_summary(dev, frame, steps=args.steps, warmup=args.warmup, arith=args.arith)
    out.update(metric="train_step_ms", points=args.points)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
