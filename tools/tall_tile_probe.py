#!/usr/bin/env python
"""GPU tool: the recorded contractions of a stack-mode batch-16 forward with <= 64 output columns, default plan against the forced
128 x 64 tile (bf16x6): outputs bit-equal?  time per launch?    python tools/tall_tile_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tools.gemm_shapes import time_graph
from tools.tune_gemm import shape_of


def main():
    from cofii2p_amd import _lib, ops
    from cofii2p_amd.network import CoFiI2P

    ops.GEMM_MODE = "bf16x6"
    lib = _lib.load()
    force_plan = lib.cofi_tune_force_plan
    force_plan.argtypes, force_plan.restype = [ctypes.c_int] * 3, ctypes.c_int
    dev = torch.device("cuda", 0)
    model = CoFiI2P(bench.Opt()).to(dev)
    frames = bench.make_inputs(dev, [0, 1], 20480)
    bench.one_step(model, frames[0])
    bsz = int(os.environ.get("PROBE_BATCH", "16"))
    grp = [frames[i % len(frames)] for i in range(bsz)]
    pyr, img = CoFiI2P.stack_frames([g[0] for g in grp], [g[1] for g in grp])
    P = model._pack(dev)
    kt = bench.KernelTimer()
    kt.record_fn(lambda: model._run_device(P, pyr["points"], pyr["neighbors"], pyr["subsampling"], pyr["upsampling"], pyr["feats"], img, "test", None, None))
    seen = {}
    for name in ("gemm", "gemm_colstats", "gemm_layernorm", "conv2d_nhwc"):
        for fn, a, k, _ in kt.calls.get(name, []):
            sh = shape_of(name, a, k)
            if sh[1] <= int(os.environ.get("PROBE_MAXN", "64")) and sh[1] > int(os.environ.get("PROBE_MINN", "32")) and sh[0] >= 8192:
                seen.setdefault((name,) + sh, [fn, a, k, 0])[3] += 1
    tot0 = tot1 = 0.0
    for key, (fn, a, k, cnt) in sorted(seen.items()):
        def run():
            return fn(*a, **k)

        def outs(r):
            r = r if isinstance(r, (tuple, list)) else (r,)
            flat = []
            for t in r:
                if torch.is_tensor(t):
                    flat.append(t.clone())
                elif hasattr(t, "part") and torch.is_tensor(getattr(t, "part", None)):
                    flat.append(t.part.clone())
            return flat

        force_plan(0, 0, 0)
        r0 = outs(run())
        t0 = time_graph(run, reps=10) * 1e6
        force_plan(int(os.environ.get("PROBE_BM", "128")), 64, 1)
        r1 = outs(run())
        t1 = time_graph(run, reps=10) * 1e6
        force_plan(0, 0, 0)
        same = len(r0) == len(r1) and all(torch.equal(x, y) for x, y in zip(r0, r1))
        tot0 += t0 * cnt
        tot1 += t1 * cnt
        print("%-60s x%-2d default %8.2f us   forced %8.2f us   (%+5.1f %%)   outputs %s (%d tensors)" % (key, cnt, t0, t1, 100 * (t1 / t0 - 1), "bit-equal" if same else "DIFFER", len(r0)))
    print("sum over the forward: default %.1f us, forced %.1f us per submission of %d frames" % (tot0, tot1, bsz))


if __name__ == "__main__":
    main()
