#!/usr/bin/env python
"""GPU tool: the recorded 3 x 3 convolutions with 64 output channels of a stack-mode forward, implicit-GEMM plan against the direct kernel
(csrc/conv_direct.inc): time per launch, largest difference.    python tools/conv_direct_probe.py   (PROBE_BATCH=16)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tools.gemm_shapes import time_graph
from tools.tune_gemm import shape_of
from tools.big_gemm_probe import outs


def main():
    from cofii2p_amd import _lib, ops
    from cofii2p_amd.network import CoFiI2P

    ops.GEMM_MODE = "bf16x6"
    lib = _lib.load()
    hook = lib.cofi_tune_force_conv_direct
    hook.argtypes, hook.restype = [ctypes.c_int], ctypes.c_int
    dev = torch.device("cuda", 0)
    model = CoFiI2P(bench.Opt()).to(dev)
    frames = bench.make_inputs(dev, [0, 1], 20480)
    bench.one_step(model, frames[0])
    bsz = int(os.environ.get("PROBE_BATCH", "16"))
    grp = [frames[i % len(frames)] for i in range(bsz)]
    pyr, img = CoFiI2P.stack_frames([g[0] for g in grp], [g[1] for g in grp])
    P = model._pack(dev)
    kt = bench.KernelTimer()
    hook(-1)
    kt.record_fn(lambda: model._run_device(P, pyr["points"], pyr["neighbors"], pyr["subsampling"], pyr["upsampling"], pyr["feats"], img, "test", None, None))
    seen = {}
    for fn, a, k, (fl, by) in kt.calls.get("conv2d_nhwc", []):
        sh = shape_of("conv2d_nhwc", a, k)
        if sh[1] == 64 and sh[2] % 9 == 0:
            seen.setdefault(tuple(sh), [fn, a, k, 0, fl])[3] += 1
    t_old = t_new = 0.0
    for key, (fn, a, k, cnt, fl) in sorted(seen.items()):
        run = lambda: fn(*a, **k)
        hook(-1)
        r0 = outs(run())
        t0 = time_graph(run, reps=6) * 1e6
        hook(1)
        r1 = outs(run())
        t1 = time_graph(run, reps=6) * 1e6
        hook(0)
        diff = max(float((x - y).abs().max() / max(1e-30, float(y.abs().max()))) for x, y in zip(r1, r0))
        t_old += t0 * cnt
        t_new += t1 * cnt
        print("%-26s x%-2d implicit %8.1f us (%5.1f TF/s)  direct %8.1f us (%5.1f TF/s)  max relative difference %.2e" % (
            key, cnt, t0, fl / t0 * 1e-6, t1, fl / t1 * 1e-6, diff))
    print("sum: implicit %.1f us, direct %.1f us per submission of %d frames" % (t_old, t_new, bsz))


if __name__ == "__main__":
    main()
