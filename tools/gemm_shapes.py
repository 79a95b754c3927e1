#!/usr/bin/env python
"""Per-shape timing of every cofi_gemm_f32 launch of one KITTI frame (GPU tool, not a test).
    python tools/gemm_shapes.py [--points 20480]"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def time_graph(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20480)
    ap.add_argument("--kernel", default="gemm")
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    from cofii2p_amd.network import CoFiI2P

    dev = torch.device("cuda", 0)
    model = CoFiI2P(bench.Opt()).to(dev)
    kt = bench.record_kernel_calls(model, dev, args.points, args.batch)
    rows = {}
    names = ["gemm", "gemm_colstats", "gemm_layernorm", "conv2d_nhwc"] if args.kernel == "gemm" else [args.kernel]
    calls = [(n,) + c for n in names for c in kt.calls.get(n, [])]
    for name, fn, a, k, (fl, by) in calls:
        if name == "conv2d_nhwc":
            key = ("conv", a[0].shape[0], a[3].shape[0], a[3].shape[1], "s%d" % (a[5] if len(a) > 5 else 1))
        elif args.kernel == "gemm":
            key = (name.replace("gemm", "g"), a[0].shape[0], a[1].shape[0], a[0].shape[1], "div" if k.get("rowdiv") is not None else "")
        else:
            key = tuple(tuple(t.shape) for t in a if torch.is_tensor(t))[:4]
        t = time_graph(lambda: fn(*a, **k))
        r = rows.setdefault(key, [0, 0.0, fl, by])
        r[0] += 1
        r[1] += t
    tot = sum(r[1] for r in rows.values())
    print("%-46s %5s %9s %9s %8s %9s" % ("shape", "calls", "us/call", "TF/s", "% time", "GB/s (algorithmic)"))
    for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%-46s %5d %9.2f %9.2f %8.1f %9.1f" % (str(key), r[0], 1e6 * r[1] / r[0], r[2] / (r[1] / r[0]) / 1e12, 100 * r[1] / tot, r[3] / (r[1] / r[0]) / 1e9))
    print("total %.1f us per submission (batch %d: %.1f us per frame)" % (1e6 * tot, args.batch, 1e6 * tot / args.batch))


if __name__ == "__main__":
    main()
