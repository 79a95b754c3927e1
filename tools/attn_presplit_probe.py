#!/usr/bin/env python
"""GPU tool: the bf16x6 attention kernel with K / V split by every workgroup (cofi_attention_parts_bf16x6) against K / V split once
(cofi_attention_kv_planes + cofi_attention_parts_planes) on the launch shapes of the bench configurations: us per launch (hipGraph replay),
the split launch alone, bit equality of the merged outputs.
    python tools/attn_presplit_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.attn_probe import SHAPES
from tools.gemm_shapes import time_graph


def main():
    from cofii2p_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    ops.ATTN_MODE = "bf16x6"
    print("%-42s %12s %12s %12s %12s  %s" % ("launch", "in-kernel us", "split us", "kernel us", "both us", "TF/s of 416.7 (in-kernel -> planes kernel alone / both launches)"))
    for name, frames, L, S in SHAPES:
        q = torch.randn(frames * L, 128, generator=g).to(dev)
        kv = (torch.randn(frames * S, 256, generator=g) * 1.5).to(dev)
        k, v = kv[:, :128], kv[:, 128:]
        cs = (torch.rand(frames, 128, generator=g) + 0.5).to(dev)
        flop = 4.0 * frames * L * S * 128
        ops.ATTN_PRESPLIT_ROWS = 0
        ref = ops.attention(q, k, v, q_colscale=cs, frames=frames).clone()
        t0 = min(time_graph(lambda: ops.attention_parts(q, k, v, q_colscale=cs, frames=frames), reps=10) for _ in range(2))
        ops.ATTN_PRESPLIT_ROWS = 1
        out = ops.attention(q, k, v, q_colscale=cs, frames=frames).clone()
        tb = min(time_graph(lambda: ops.attention_parts(q, k, v, q_colscale=cs, frames=frames), reps=10) for _ in range(2))
        pb = lib.cofi_attention_kv_planes_bytes(S, 4, 32, frames)
        img = torch.empty(pb, dtype=torch.uint8, device=dev)

        def split():
            _lib.check(lib.cofi_attention_kv_planes(ops._p(k), ops._ld(k), ops._p(v), ops._ld(v), S, 4, 32, frames, ops._p(img), img.numel(), ops._stream()), "kv_planes")

        ts = min(time_graph(split, reps=10) for _ in range(2))
        tf = lambda t: flop / t * 1e-12
        print("%-42s %12.1f %12.1f %12.1f %12.1f  %.3f -> %.3f / %.3f   %s" % (name, t0 * 1e6, ts * 1e6, (tb - ts) * 1e6, tb * 1e6, tf(t0) / 416.7, tf(tb - ts) / 416.7, tf(tb) / 416.7,
                                                                             "bit-equal" if torch.equal(ref, out) else "DIFFER %.3e" % float((ref - out).abs().max())))
    ops.ATTN_MODE = "auto"


if __name__ == "__main__":
    main()
