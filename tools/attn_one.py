#!/usr/bin/env python
"""GPU tool: ONE attention launch shape, repeated (profiling target for rocprofv3 passes).
    python tools/attn_one.py --frames 16 --L 1280 --S 1280 --arith bf16x6 --reps 5"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--L", type=int, default=1280)
    ap.add_argument("--S", type=int, default=1280)
    ap.add_argument("--arith", default="bf16x6", choices=["bf16x6", "f32"])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from cofii2p_amd import _lib, ops

    _lib.load()
    ops.ATTN_MODE = args.arith
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(args.frames * args.L, 128, generator=g, device=dev)
    k = torch.randn(args.frames * args.S, 128, generator=g, device=dev) * 2
    v = torch.randn(args.frames * args.S, 128, generator=g, device=dev)
    for _ in range(args.reps):
        ops.attention_parts(q, k, v, frames=args.frames)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
