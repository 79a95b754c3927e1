#!/usr/bin/env python
"""GPU tool: the recorded contractions of a stack-mode forward that the 256 x 128 bf16x6 kernel (csrc/gemm_x6_big.inc) can take,
small-tile plan against the big kernel at several K splits: outputs bit-equal at equal split?  time per launch?
    python tools/big_gemm_probe.py            (PROBE_BATCH=16, PROBE_KS="0,1,2,3,4", PROBE_MINK=256)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tools.gemm_shapes import time_graph
from tools.tune_gemm import shape_of


def outs(r):
    r = r if isinstance(r, (tuple, list)) else (r,)
    flat = []
    for t in r:
        if torch.is_tensor(t):
            flat.append(t.clone())
        elif hasattr(t, "part") and torch.is_tensor(getattr(t, "part", None)):
            flat.append(t.part.clone())
    return flat


def main():
    from cofii2p_amd import _lib, ops
    from cofii2p_amd.network import CoFiI2P

    ops.GEMM_MODE = "bf16x6"
    lib = _lib.load()
    force_plan = lib.cofi_tune_force_plan
    force_plan.argtypes, force_plan.restype = [ctypes.c_int] * 3, ctypes.c_int
    force_big = lib.cofi_tune_force_big
    force_big.argtypes, force_big.restype = [ctypes.c_int] * 2, ctypes.c_int
    dev = torch.device("cuda", 0)
    npts = 20480
    if os.environ.get("PROBE_STRESS", "0") == "1":   # BASELINE configs[4]: 896 x 1600 image, 40960 points
        bench.Opt.img_H, bench.Opt.img_W, npts = 896, 1600, 40960
    model = CoFiI2P(bench.Opt()).to(dev)
    frames = bench.make_inputs(dev, [0, 1], npts)
    bench.one_step(model, frames[0])
    bsz = int(os.environ.get("PROBE_BATCH", "16"))
    ks_list = [int(x) for x in os.environ.get("PROBE_KS", "0,1,2,3,4").split(",")]
    mink = int(os.environ.get("PROBE_MINK", "256"))
    grp = [frames[i % len(frames)] for i in range(bsz)]
    pyr, img = CoFiI2P.stack_frames([g[0] for g in grp], [g[1] for g in grp])
    P = model._pack(dev)
    kt = bench.KernelTimer()
    force_big(-1, 0)
    kt.record_fn(lambda: model._run_device(P, pyr["points"], pyr["neighbors"], pyr["subsampling"], pyr["upsampling"], pyr["feats"], img, "test", None, None))
    seen = {}
    for name in ("gemm", "gemm_colstats", "conv2d_nhwc"):
        for fn, a, k, (fl, by) in kt.calls.get(name, []):
            sh = shape_of(name, a, k)   # (M, N, K)
            if int(os.environ.get("PROBE_MINN", "128")) <= sh[1] <= int(os.environ.get("PROBE_MAXN", "100000")) and sh[2] >= mink and sh[2] % 32 == 0 and sh[0] >= int(os.environ.get("PROBE_MINM", "4096")):
                seen.setdefault((name,) + tuple(sh), [fn, a, k, 0, fl])[3] += 1
    tot = {}
    print("%-44s %3s %9s | %s" % ("shape", "x", "small us", "  ".join("big ks=%d" % k for k in ks_list)))
    for key, (fn, a, k, cnt, fl) in sorted(seen.items(), key=lambda kv: -kv[1][4] * kv[1][3]):
        def run():
            return fn(*a, **k)

        force_big(-1, 0)
        t0 = time_graph(run, reps=6) * 1e6
        # bit-equality at equal split (ks = 1): the 128 x 128 small-tile plan against the big kernel
        force_plan(128, 128 if key[2] > 64 else 64, 1)
        r0 = outs(run())
        force_plan(0, 0, 0)
        force_big(1, 1)
        r1 = outs(run())
        same = len(r0) == len(r1) and all(torch.equal(x, y) for x, y in zip(r0, r1))
        cells = []
        best = t0
        for ks in ks_list:
            force_big(1, ks)
            t1 = time_graph(run, reps=6) * 1e6
            best = min(best, t1)
            cells.append("%8.1f" % t1)
        force_big(0, 0)
        tauto = time_graph(run, reps=6) * 1e6
        force_big(-1, 0)
        tot.setdefault("small", 0.0)
        tot["small"] += t0 * cnt
        tot.setdefault("best", 0.0)
        tot["best"] += best * cnt
        tot.setdefault("auto", 0.0)
        tot["auto"] += tauto * cnt
        print("%-44s x%-2d %9.1f | %s | shipped plan %8.1f | %5.1f TF/s -> %5.1f  %s" % (
            key, cnt, t0, "  ".join(cells), tauto, fl / t0 * 1e-6, fl / best * 1e-6, "bit-equal" if same else "DIFFER"))
    force_big(0, 0)
    print("sum over the submission (%d frames): small-tile %.1f us, best of the probe %.1f us, shipped plans %.1f us" % (bsz, tot["small"], tot["best"], tot["auto"]))


if __name__ == "__main__":
    main()
