#!/usr/bin/env python
"""GPU tool: where a launch of the bf16x6 attention kernel spends its time - the kernel rebuilt WITHOUT one part at a time
(csrc/attention_x6.inc DBG: the results of those builds are wrong, only their duration means something).
Needs a library built with the ablation instantiations:  COFI_HIPCC_FLAGS=-DCOFI_ATTN_ABLATION python -m cofii2p_amd.build --force
    python tools/attn_ablate.py [frames L S]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.gemm_shapes import time_graph

PARTS = [(0, "complete kernel"), (1, "no global loads in the loop"), (2, "no barriers in the loop"), (3, "no loads, no barriers"), (4, "no softmax"),
         (8, "no staging split / LDS writes"), (9, "no loads, no staging split"), (11, "no loads, no barriers, no staging split"),
         (15, "... and no softmax (fragment reads + both chains + P split)"), (16, "no second chain (P split + 12 MFMA)"), (32, "no first chain (12 MFMA)")]


def main():
    from cofii2p_amd import _lib, ops

    lib = _lib.load()
    if not hasattr(lib, "cofi_tune_attention_x6_debug"):
        sys.exit("this libcofi_hip.so has no ablation builds: COFI_HIPCC_FLAGS=-DCOFI_ATTN_ABLATION python -m cofii2p_amd.build --force")
    dbg = lib.cofi_tune_attention_x6_debug
    dbg.argtypes, dbg.restype = [ctypes.c_int], ctypes.c_int
    frames, L, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 1280, 1280)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(frames * L, 128, generator=g, device=dev)
    k = torch.randn(frames * S, 128, generator=g, device=dev) * 2
    v = torch.randn(frames * S, 128, generator=g, device=dev)
    ops.ATTN_MODE = "bf16x6"
    print("attention_x6_kernel, frames %d, L %d, S %d" % (frames, L, S))
    for flags, what in PARTS:
        dbg(flags)
        t = time_graph(lambda: ops.attention_parts(q, k, v, frames=frames), reps=10)
        print("  %-70s %8.1f us" % (what, t * 1e6))
    dbg(0)


if __name__ == "__main__":
    main()
