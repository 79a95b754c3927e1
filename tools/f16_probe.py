#!/usr/bin/env python
"""GPU tool: the recorded contractions of a stack-mode forward that take the 256 x 128 kernel, six-product bf16 split (gemm_x6_big_kernel)
against the three-product fp16 split (gemm_f16_big_kernel + its repair launch) ON THE FORWARD'S OWN OPERANDS: time per launch (hipGraph
replay, alternating), largest deviation between the two results relative to the output's rms, repair events.
    python tools/f16_probe.py            (PROBE_BATCH=16)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tools.big_gemm_probe import outs
from tools.gemm_shapes import time_graph
from tools.tune_gemm import shape_of


def main():
    from cofii2p_amd import _lib, ops
    from cofii2p_amd.network import CoFiI2P

    ops.GEMM_MODE = "bf16x6"
    lib = _lib.load()
    ev = lib.cofi_tune_f16x3_resplit_events
    ev.argtypes, ev.restype = [ctypes.c_int], ctypes.c_long
    lib.cofi_tune_big_debug.argtypes, lib.cofi_tune_big_debug.restype = [ctypes.c_int], ctypes.c_int
    lib.cofi_tune_big_debug(int(os.environ.get("PROBE_DBG", "0")))   # 256: the four-wave geometry of the f16x3 kernel
    dev = torch.device("cuda", 0)
    model = CoFiI2P(bench.Opt()).to(dev)
    frames = bench.make_inputs(dev, [0, 1], 20480)
    bench.one_step(model, frames[0])
    bsz = int(os.environ.get("PROBE_BATCH", "16"))
    grp = [frames[i % len(frames)] for i in range(bsz)]
    pyr, img = CoFiI2P.stack_frames([g[0] for g in grp], [g[1] for g in grp])
    P = model._pack(dev)
    kt = bench.KernelTimer()
    ops.F16X3_BIG = False
    kt.record_fn(lambda: model._run_device(P, pyr["points"], pyr["neighbors"], pyr["subsampling"], pyr["upsampling"], pyr["feats"], img, "test", None, None))
    seen = {}
    for name in ("gemm", "gemm_colstats", "conv2d_nhwc"):
        for fn, a, k, (fl, by) in kt.calls.get(name, []):
            sh = shape_of(name, a, k)   # (M, N, K)
            if sh[1] >= 128 and sh[2] >= 256 and sh[2] % 32 == 0 and sh[0] >= 4096:
                seen.setdefault((name,) + tuple(sh), [fn, a, k, 0, fl])[3] += 1
    tot = {"x6": 0.0, "f16": 0.0}
    print("%-44s %3s %10s %10s %7s %9s %9s %s" % ("shape", "x", "bf16x6 us", "f16x3 us", "ratio", "TF/s x6", "TF/s f16", "max |diff| / rms(out), repairs"))
    for key, (fn, a, k, cnt, fl) in sorted(seen.items(), key=lambda kv: -kv[1][4] * kv[1][3]):
        def run():
            return fn(*a, **k)

        ops.F16X3_BIG = False
        r0 = outs(run())
        t0 = time_graph(run, reps=6) * 1e6
        ops.F16X3_BIG = True
        ev(1)
        r1 = outs(run())
        torch.cuda.synchronize()
        nev = ev(1)
        t1 = time_graph(run, reps=6) * 1e6
        ops.F16X3_BIG = False
        t0b = time_graph(run, reps=6) * 1e6
        ops.F16X3_BIG = True
        t1b = time_graph(run, reps=6) * 1e6
        t0, t1 = min(t0, t0b), min(t1, t1b)
        dev_ = max(float((x.double() - y.double()).abs().max() / (y.double().pow(2).mean().sqrt() + 1e-30)) for x, y in zip(r1[:1], r0[:1]))
        same = torch.equal(r0[0], r1[0])
        tot["x6"] += t0 * cnt
        tot["f16"] += t1 * cnt
        print("%-44s x%-2d %10.1f %10.1f %7.2f %9.1f %9.1f  %.2e  %d%s" % (key, cnt, t0, t1, t0 / t1, fl / t0 * 1e-6, fl / t1 * 1e-6, dev_, nev,
                                                                         "  (SAME BITS: the f16x3 kernel did not run)" if same else ""))
    ops.F16X3_BIG = True
    print("sum over the submission (%d frames): bf16x6 %.1f us, f16x3 %.1f us (%.2f x)" % (bsz, tot["x6"], tot["f16"], tot["x6"] / max(tot["f16"], 1e-9)))


if __name__ == "__main__":
    main()
