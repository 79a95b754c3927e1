#!/usr/bin/env python
"""GPU tool: the attention kernel (cofi_attention_parts, partial slots only - no merge) in both arithmetics on the launch shapes of
the bench configurations: the exact fp32 matrix instruction (csrc/attention.hip, roof 157.3 TF/s) against the fp32-grade bf16 split
(csrc/attention_x6.inc, roof 2500 / 6 = 416.7 TF/s); difference of the merged outputs.
    python tools/attn_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.gemm_shapes import time_graph

SHAPES = [  # name, frames, L, S
    ("batch 16 cross (image <-> points)", 16, 1280, 1280),
    ("batch 16 joint self (2 x 16 sequences)", 32, 1280, 1280),
    ("batch 1 cross", 1, 1280, 1280),
    ("batch 1 joint self", 2, 1280, 1280),
    ("stress cross image <- points", 1, 22400, 2560),
    ("stress cross points <- image", 1, 2560, 22400),
    ("stress self image", 1, 22400, 22400),
]


def main():
    from cofii2p_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    print("%-42s %10s | %s" % ("launch", "fp32 us", "bf16x6 us (TF/s, of 416.7), twice"))
    for name, frames, L, S in SHAPES:
        q = torch.randn(frames * L, 128, generator=g).to(dev)
        k = (torch.randn(frames * S, 128, generator=g) * 2).to(dev)
        v = torch.randn(frames * S, 128, generator=g).to(dev)
        cs = (torch.rand(frames, 128, generator=g) + 0.5).to(dev)
        flop = 4.0 * frames * L * S * 128
        ops.ATTN_MODE = "f32"
        ref = ops.attention(q, k, v, q_colscale=cs, frames=frames).clone()
        t0 = time_graph(lambda: ops.attention_parts(q, k, v, q_colscale=cs, frames=frames), reps=10)
        ops.ATTN_MODE = "bf16x6"
        cells, diff = [], 0.0
        for _ in range(2):   # (the second timing of a pair is usually ~4 % faster: same kernel)
            out = ops.attention(q, k, v, q_colscale=cs, frames=frames)
            diff = max(diff, float((out - ref).abs().max() / ref.abs().max()))
            t1 = time_graph(lambda: ops.attention_parts(q, k, v, q_colscale=cs, frames=frames), reps=10)
            cells.append("%8.1f (%5.1f, %.3f)" % (t1 * 1e6, flop / t1 * 1e-12, flop / t1 * 1e-12 / 416.7))
        print("%-42s %8.1f (%5.1f TF/s, %.3f of 157.3) | %s | max difference %.2e of the largest output" % (
            name, t0 * 1e6, flop / t0 * 1e-12, flop / t0 * 1e-12 / 157.3, "  ".join(cells), diff))
    ops.ATTN_MODE = "auto"


if __name__ == "__main__":
    main()
