#!/usr/bin/env python
"""bf16x6 arithmetic on the GPU box (tool): error against fp64 and time per launch of the three arithmetics on forward / backward shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cofii2p_amd import ops
from tools.planes_bench import time_graph

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(3)
shapes = [(20480, 32, 480), (20480, 64, 64), (10240, 64, 960), (5120, 128, 1920), (2560, 256, 3840), (1280, 512, 7680), (1280, 128, 128), (1280, 1024, 2048),
          (5120, 128, 1152), (200, 96, 200), (64, 480, 20480)]
print("%-22s %s" % ("shape", "  ".join("%-34s" % m for m in ("f32", "bf16x3", "bf16x6"))))
for M, N, K in shapes:
    a = torch.randn((M, K), device=dev, generator=gen)
    w = torch.randn((N, K), device=dev, generator=gen) / K ** 0.5
    bias = torch.randn((N,), device=dev, generator=gen)
    ref = a.double() @ w.double().t() + bias.double()
    sc = float(ref.abs().max())
    row = []
    for mode in ("f32", "bf16x3", "bf16x6"):
        ops.GEMM_MODE = mode
        y = ops.gemm(a, w, bias=bias)
        torch.cuda.synchronize()
        err = float((y.double() - ref).abs().max()) / sc
        rms = float((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        t = time_graph(lambda: ops.gemm(a, w, bias=bias), reps=10)
        row.append("max %.1e rms %.1e %7.1f us" % (err, rms, 1e6 * t))
    print("%-22s %s" % ((M, N, K), "  ".join("%-34s" % r for r in row)), flush=True)
# a convolution (implicit GEMM path)
H, W, Cin, Cout = 40, 128, 64, 64
x = torch.randn((H * W, Cin), device=dev, generator=gen)
wt = torch.randn((Cout, 9 * Cin), device=dev, generator=gen) / (9 * Cin) ** 0.5
ref = torch.nn.functional.conv2d(x.t().reshape(1, Cin, H, W).double(), wt.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double(), padding=1)[0].reshape(Cout, -1).t()
for mode in ("f32", "bf16x3", "bf16x6"):
    ops.GEMM_MODE = mode
    y = ops.conv2d_nhwc(x, H, W, wt, 3)[0]
    print("conv 3x3 %s: max rel err %.1e" % (mode, float((y.double() - ref).abs().max() / ref.abs().max())))
