#!/usr/bin/env python
"""Per-stage wall time of the device-side loader (cofii2p_amd/dataside.py), GPU tool:  python tools/dataside_time.py"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cofii2p_amd import dataside, synth
from cofii2p_amd.preprocess import build_pyramid

dev = torch.device("cuda", 0)
opt = types.SimpleNamespace(img_H=160, img_W=512, num_pc=20480, num_kpt=64, P_tx_amplitude=10, P_ty_amplitude=0, P_tz_amplitude=10, P_Rx_amplitude=0.0,
                            P_Ry_amplitude=2 * np.pi, P_Rz_amplitude=0.0)
raw, img, K = synth.make_raw_scan(0)
cal = dataside.calib_matrices(synth.KITTI_CALIB_LINES)
P_Tr = np.dot(cal["P2"], cal["Tr"])
raw_d, img_d, Ptr_d = torch.from_numpy(raw).to(dev), torch.from_numpy(img).to(dev), torch.from_numpy(P_Tr).to(dev)
prep = dataside.FramePreparer(opt, dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


t_vox, (vox, nvox) = timed(lambda: prep.voxel_downsample(raw_d, Ptr_d))
s = dataside.FrameSampler(0)
choice = s.downsample_choice(nvox, opt.num_pc)
P = s.random_transform(opt)
t_gather, (pts, feats) = timed(lambda: prep.resample_transform(vox, choice, P))
sub = [torch.from_numpy(i).to(dev) for i in s.subsample_indices(opt.num_pc)]
t_pyr, pyr = timed(lambda: build_pyramid(pts, sub, int64=True))
K_2, K_4, crop, rhw = dataside.intrinsics_and_crop(K, img.shape[:2], opt, s)
t_img, _ = timed(lambda: prep.image(img_d, rhw, crop))
t_draw, _ = timed(lambda: (dataside.FrameSampler(0).downsample_choice(nvox, opt.num_pc), dataside.FrameSampler(0).subsample_indices(opt.num_pc)))
t_lab, _ = timed(lambda: dataside.project_labels(pyr["points"][-1].cpu().numpy(), P, K_2, K_4, opt, dataside.FrameSampler(0)))
t_all, _ = timed(lambda: prep.prepare(raw_d, img_d, K, P_Tr, 0))
print("raw points %d -> voxels %d" % (raw.shape[1], nvox))
for name, t in (("voxel grid (pack + bounds + keys + 5 radix passes x 3 launches + heads + means, incl. the count sync)", t_vox), ("gather + SE(3) (incl. H2D of choice, P)", t_gather),
                ("KNN pyramid (int64 tables)", t_pyr), ("image resize + crop", t_img), ("host draws (choice + sub-sampling)", t_draw),
                ("labels (D2H coarse points + numpy)", t_lab), ("prepare() end to end", t_all)):
    print("%8.3f ms  %s" % (t, name))
