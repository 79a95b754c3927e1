#!/usr/bin/env python
"""Replay every call of ONE C-ABI entry point of a KITTI frame `reps` times, eagerly (profiling target for
rocprofv3 --pmc passes: the trace then only contains the kernels of interest plus one set-up forward).
    python tools/replay_kernel.py --kernel kpconv_aggregate --reps 5"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="kpconv_aggregate")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--points", type=int, default=20480)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    from cofii2p_amd.network import CoFiI2P

    dev = torch.device("cuda", 0)
    model = CoFiI2P(bench.Opt()).to(dev)
    kt = bench.record_kernel_calls(model, dev, args.points, args.batch)
    calls = kt.calls.get(args.kernel, [])
    torch.cuda.synchronize()
    for _ in range(args.reps):
        for fn, a, k, _w in calls:
            fn(*a, **k)
    torch.cuda.synchronize()
    print("replayed %d calls x %d" % (len(calls), args.reps))


if __name__ == "__main__":
    main()
