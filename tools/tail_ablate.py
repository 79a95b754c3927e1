#!/usr/bin/env python
"""Timing of one fused-layer-tail launch (GPU tool); COFI_TAIL_ABLATE selects an ablated kernel; argv[1] = rows (1280), argv[2] = 'parts'|'plain'."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cofii2p_amd import ops, transformer
from tools.gemm_shapes import time_graph

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
mode = sys.argv[2] if len(sys.argv) > 2 else "parts"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
sd = {n + ".weight": (torch.randn(*shp, generator=g) / shp[1] ** 0.5).to(dev) for n, shp in
      (("q_proj", (128, 128)), ("k_proj", (128, 128)), ("v_proj", (128, 128)), ("merge", (128, 128)), ("mlp.0", (256, 256)), ("mlp.2", (128, 256)))}
for n in ("norm1", "norm2"):
    sd[n + ".weight"], sd[n + ".bias"] = torch.ones(128, device=dev), torch.zeros(128, device=dev)
w = transformer.pack_layer(sd, "")
x = torch.randn(L, 128, generator=g).to(dev)
q, k, v = (torch.randn(L, 128, generator=g).to(dev) for _ in range(3))
out = torch.empty_like(x)
msg = ops.attention(q, k, v, parts=(mode == "parts"))
t = time_graph(lambda: ops.loftr_tail(msg, x, w, out), reps=20)
print("tail ablate=%s rows=%d msg=%s: %.2f us" % (os.environ.get("COFI_TAIL_ABLATE", "0"), L, mode, 1e6 * t))
