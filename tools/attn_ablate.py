#!/usr/bin/env python
"""Timing of one attention launch (GPU tool): python tools/attn_ablate.py [L S frames]; COFI_ATTN_ABLATE selects an ablated kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cofii2p_amd import ops
from tools.gemm_shapes import time_graph

L, S, frames = (int(x) for x in (sys.argv[1:4] + ["1280", "1280", "1"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
q = torch.randn(frames * L, 128, generator=g).to(dev)
k = torch.randn(frames * S, 128, generator=g).to(dev)
v = torch.randn(frames * S, 128, generator=g).to(dev)
t = time_graph(lambda: ops.attention(q, k, v, frames=frames, parts=True), reps=20)
fl = 4.0 * frames * L * S * 128
print("ablate=%s L=%d S=%d frames=%d: %.2f us  %.1f TF/s  frac %.3f" % (os.environ.get("COFI_ATTN_ABLATE", "0"), L, S, frames, 1e6 * t, fl / t / 1e12, fl / t / 157.3e12))
