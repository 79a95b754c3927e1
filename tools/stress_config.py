#!/usr/bin/env python
"""BASELINE.json configs[4]: "Nuscenes-shape (900x1600 img, 40960 pts), 1 GPU — large-attention / HBM-bound stress config".
900 is not divisible by 8/32 (the reference's own up-sampler would fail), so the image is 896x1600.
Runs the forward in test mode, checks finiteness / shapes, prints timing.  GPU tool."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    from cofii2p_amd import ops
    from cofii2p_amd.network import CoFiI2P

    H, W, NP = 896, 1600, 40960

    class Opt:
        img_H, img_W, img_fine_resolution_scale, norm = H, W, 32, "gn"

    ops.GEMM_MODE = "bf16x3"
    dev = torch.device("cuda", 0)
    model = CoFiI2P(Opt()).to(dev)
    bench.Opt.img_H, bench.Opt.img_W = H, W
    frames = bench.make_inputs(dev, [0], NP)
    pyr, img, _ = frames[0]
    torch.cuda.synchronize()
    out = model(pyr, img, None, None, None, "test")
    torch.cuda.synchronize()
    shapes = [tuple(t.shape) for t in out]
    assert all(torch.isfinite(t).all() for t in out)
    model.enable_graphs()
    for _ in range(2):
        model(pyr, img, None, None, None, "test")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        model(pyr, img, None, None, None, "test")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    att_flops = 4.0 * 128 * (4 * 2 * 22400.0 * 22400 + 4 * 2 * 2560.0 * 2560 + 4 * 2 * 2 * 22400.0 * 2560) / 2
    print(json.dumps({"config": "stress 896x1600 image (22400 tokens), 40960 points", "ms_per_frame": 1e3 * dt, "frames_per_s": 1 / dt,
                      "output_shapes": shapes, "matches": shapes[4][0], "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
