#!/usr/bin/env python
"""GPU box experiment: the headline pipeline (4 frame streams) next to a FIFTH stream user - a loader / RCCL-like stream that enqueues a
small copy per frame - and four FRESH frame streams instead of the process's own (capture stream + default stream first).
    [GPU_MAX_HW_QUEUES=8] python tools/fifth_stream.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from cofii2p_amd.network import CoFiI2P

dev = torch.device("cuda", 0)
model = CoFiI2P(bench.Opt()).to(dev)
model.enable_graphs(True)
frames = bench.make_inputs(dev, list(range(8)), 20480)


def rate(streams, extra=None, n=200):
    NSL = 2 * len(streams)
    pend = [None] * NSL
    src, dst = torch.zeros(1 << 16, device=dev), torch.zeros(1 << 16, device=dev)
    for phase in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            sl = i % NSL
            if pend[sl] is not None:
                model.finish(pend[sl])
            pyr, img, _ = frames[i % len(frames)]
            with torch.cuda.stream(streams[i % len(streams)]):
                pend[sl] = model.forward_async(sl, pyr, img, inputs_stable=True)
            if extra is not None:
                with torch.cuda.stream(extra):
                    dst.copy_(src, non_blocking=True)
        for sl in range(NSL):
            if pend[sl] is not None:
                model.finish(pend[sl])
                pend[sl] = None
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return n / dt


own = model.frame_streams(4)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default 4)"))
print("4 frame streams (the process's own first):            %.1f frames/s" % rate(own))
print("... + a fifth stream with one small copy per frame:   %.1f frames/s" % rate(own, torch.cuda.Stream(device=dev)))
fresh = [torch.cuda.Stream(device=dev) for _ in range(4)]
print("4 FRESH frame streams (default + capture stream idle): %.1f frames/s" % rate(fresh))
