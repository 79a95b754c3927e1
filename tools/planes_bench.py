#!/usr/bin/env python
"""gemm_planes_kernel on the GPU box (tool, not a test): bit-identity against the register-staged bf16x3 kernel at equal split-K,
then timing of every configuration x split per shape; writes the winners as cofii2p_amd/csrc/gemm_planes_plans.inc rows.
    python tools/planes_bench.py [--quick] [--out gpurun_out/gemm_planes_plans.inc] [--batches 1,16]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cofii2p_amd import _lib, ops

CFGS = [(128, 128, 64, 2), (64, 64, 64, 2), (256, 128, 32, 3), (256, 256, 32, 2), (128, 32, 64, 2), (128, 64, 64, 2), (64, 64, 32, 4), (128, 128, 32, 4),
        (64, 128, 64, 2)]   # (bm, bn, bk, stages) by configuration id: kPlanesCfg of gemm.hip


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
    best = 1e9
    for _ in range(3):
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e-3 / reps)
    return best


def split_a(a):
    """fp32 (M, K) -> ops.SplitA with the library's own rounding (cofi_split_bf16_planes)."""
    w = ops.SplitW(a)
    return ops.SplitA(w.planes, a.shape[1])


def kpconv_shapes(batches):
    base = [(20480, 32, 480), (10240, 32, 480), (10240, 64, 960), (5120, 64, 960), (5120, 128, 1920), (2560, 128, 1920), (2560, 256, 3840),
            (1280, 256, 3840), (1280, 512, 7680)]
    out = []
    for b in batches:
        for M, N, K in base:
            if (M * b, N, K) not in out:
                out.append((M * b, N, K))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/gemm_planes_plans.inc")
    ap.add_argument("--batches", default="1,16")
    ap.add_argument("--quick", action="store_true", help="correctness sweep only")
    ap.add_argument("--shapes", default="", help="extra M,N,K;M,N,K list")
    args = ap.parse_args()
    ops.GEMM_MODE = "bf16x3"
    lib = _lib.load()
    fp, fo = lib.cofi_tune_force_planes, lib.cofi_tune_force_plan
    fp.argtypes, fp.restype = [ctypes.c_int] * 2, ctypes.c_int
    fo.argtypes, fo.restype = [ctypes.c_int] * 3, ctypes.c_int
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(7)

    def problem(M, N, K):
        a = torch.randn((M, K), device=dev, generator=gen)
        w = torch.randn((N, K), device=dev, generator=gen) / K ** 0.5
        bias = torch.randn((N,), device=dev, generator=gen)
        rowdiv = torch.randint(1, 9, (M,), device=dev, generator=gen).float()
        return a, split_a(a), ops.SplitW(w), bias, rowdiv

    # ---------------------------------------------------------------- correctness: every configuration, awkward shapes, both epilogues
    bad = 0
    cases = [(200, 96, 200), (64, 32, 480), (1280, 512, 1024), (333, 130, 72), (2560, 256, 3840), (4096, 64, 960), (130, 1, 64)]
    for M, N, K in cases:
        a, sa, sw, bias, rowdiv = problem(M, N, K)
        ref64 = (a.double() @ sw.w.double().t())
        for ks in (1, 3):
            if ks > 1 and K < 512:
                continue
            fp(-2, 0)
            fo(64, 64, ks)
            sw_ = 32 if N % 32 == 0 else (2 if N % 2 == 0 else 1)
            ref, refp = ops.gemm_colstats(sa, sw, bias=bias, rowdiv=rowdiv, act=ops.ACT_LEAKY01, stat_width=sw_)
            err = float((ops.gemm(sa, sw) - ref64).abs().max() / ref64.abs().max())
            fo(0, 0, 0)
            for cid in range(len(CFGS)):
                fp(cid, ks)
                y, part = ops.gemm_colstats(sa, sw, bias=bias, rowdiv=rowdiv, act=ops.ACT_LEAKY01, stat_width=sw_)
                torch.cuda.synchronize()
                same = torch.equal(y, ref) and torch.allclose(part, refp, rtol=2e-5, atol=1e-3)
                if not same:
                    bad += 1
                    d = float((y - ref).abs().max())
                    print("MISMATCH cfg %d %s ks %d shape %s: max |diff| %.3e, stats equal %s" % (cid, CFGS[cid], ks, (M, N, K), d, torch.equal(part, refp)), flush=True)
            print("shape %-18s ks %d: bf16x3 vs fp64 rel err %.2e; all configurations compared" % ((M, N, K), ks, err), flush=True)
    fp(-1, 0)
    print("correctness: %d mismatches" % bad, flush=True)
    if args.quick:
        return 1 if bad else 0

    # ---------------------------------------------------------------- timing
    shapes = kpconv_shapes([int(b) for b in args.batches.split(",")])
    for s in args.shapes.split(";"):
        if s:
            shapes.append(tuple(int(x) for x in s.split(",")))
    rows = []
    for M, N, K in shapes:
        a, sa, sw, bias, rowdiv = problem(M, N, K)
        fl = 2.0 * M * N * K
        sw_ = 32 if N % 32 == 0 else 1
        run_planes = lambda: ops.gemm_colstats(sa, sw, bias=bias, rowdiv=rowdiv, stat_width=sw_)
        run_f32a = lambda: ops.gemm_colstats(a, sw, bias=bias, rowdiv=rowdiv, stat_width=sw_)
        fp(-2, 0)
        t_old = time_graph(run_f32a)       # the round-2 path: fp32 A split on the fly, tuned plan
        t_old_split = time_graph(run_planes)  # round-2 kernel fed with planes
        res = []
        for cid, (bm, bn, bk, nst) in enumerate(CFGS):
            if bn > 2 * N and bn > 32:
                continue
            if bm > M:
                continue
            nb = -(-M // bm) * -(-N // bn)
            for ks in (1, 2, 3, 4, 6, 8, 12, 16):
                if ks > 1 and (K // ks < 256 or nb * ks > 4096 or nb >= 1024):
                    continue
                fp(cid, ks)
                try:
                    t = time_graph(run_planes, reps=10)
                except Exception as e:
                    print("  cfg %d ks %d failed: %s" % (cid, ks, e), flush=True)
                    continue
                res.append((t, cid, ks))
        fp(-1, 0)
        res.sort()
        t, cid, ks = res[0]
        rows.append((M, N, K, cid, ks, t, t_old))
        top = "  ".join("c%d/ks%d %.1f" % (c, k, 1e6 * tt) for tt, c, k in res[:6])
        print("%6d %5d %5d  r02 %.1f us (planes-fed %.1f) -> %.1f us = %.0f TF/s (%.2f of 833)  | %s" % (
            M, N, K, 1e6 * t_old, 1e6 * t_old_split, 1e6 * t, fl / t / 1e12, fl / t / 833.3e12, top), flush=True)
        by_cfg = {}
        for tt, c, k in res:
            by_cfg.setdefault(c, (tt, k))
        print("         per configuration: " + "  ".join("c%d %.1f(ks%d)" % (c, 1e6 * v[0], v[1]) for c, v in sorted(by_cfg.items())), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("// generated by tools/planes_bench.py on MI355X - plans of gemm_planes_kernel {M, N, K, configuration, ksplit}\n")
        f.write("static const TunedPlanes kTunedPlanes[] = {\n")
        for M, N, K, cid, ks, t, t_old in rows:
            f.write("    {%d, %d, %d, %d, %d},  // %.1f us (round-2 kernel %.1f)\n" % (M, N, K, cid, ks, 1e6 * t, 1e6 * t_old))
        f.write("};\n")
    print("saved", args.out)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
