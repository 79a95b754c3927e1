#!/usr/bin/env python
"""Diagnostic (tool, not a test): per-parameter gradient error of one train step against the fp64 gradients of tests/golden/train_ref.npz.
    python tools/train_diag.py [f32|bf16x3]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

from common import frame_inputs, load_golden
from cofii2p_amd import ops
from cofii2p_amd.network import CoFiI2P
from cofii2p_amd.train_step import step_losses

DEV = "cuda:0"
arith = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.GEMM_MODE = arith
gold = load_golden("train_ref.npz")


class Opt:
    img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"


class StepOpt:
    dist_thres, pos_margin, neg_margin = float(gold["dist_thres"]), float(gold["pos_margin"]), float(gold["neg_margin"])


fr, data = frame_inputs(int(gold["frame_id"]), int(gold["num_points"]), int(gold["pyr_seed"]))
dd = {k: [t.to(DEV) for t in v] for k, v in data.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
dd["feats"] = data["feats"].to(DEV)
img = torch.from_numpy(fr.img)[None].to(DEV)
batch = {k[4:]: torch.from_numpy(gold[k]).to(DEV) for k in gold.files if k.startswith("lab_")}
m = CoFiI2P(Opt(), arithmetic=arith).to(DEV)
m.train()
outs, mask, losses = step_losses(m, dd, img, batch, StepOpt)
for n_, t in zip(("img_desc", "pc_desc", "img_score", "pc_score", "patches", "fine_pc"), outs[:6]):
    print("output %-10s max |diff| %.3e" % (n_, float((t.detach().cpu() - torch.from_numpy(gold["train_" + n_])).abs().max())))
for name, val in zip(("loss_desc", "loss_coarse", "loss_fine"), losses):
    print("%s %.6f (reference %.6f)" % (name, float(val.detach()), float(gold[name])))
sum(losses).backward()
rows = []
params = dict(m.named_parameters())
for i, name in enumerate(str(n) for n in gold["g_names"]):
    p = params[name]
    if not int(gold["g_has"][i]):
        if p.grad is not None:
            rows.append((9.0, 9.0, name, "HAS GRAD, reference none; |g| %.3e" % float(p.grad.norm())))
        continue
    if p.grad is None:
        rows.append((9.0, 9.0, name, "NO GRAD"))
        continue
    flat = p.grad.detach().double().reshape(-1).cpu()
    norm_ref = float(gold["g_norm64"][i])
    got = flat[torch.from_numpy(gold["g_pos"][i])].numpy()
    ref = gold["g_val64"][i]
    scale = max(np.linalg.norm(ref), norm_ref * math.sqrt(len(ref) / flat.numel()), 1e-30)
    rows.append((float(np.linalg.norm(got - ref) / scale), abs(float(flat.norm()) - norm_ref) / max(norm_ref, 1e-30), name,
                 "|g| %.3e  reference fp32 vs fp64 %.3e" % (norm_ref, float(gold["g_err32"][i]))))
rows.sort(reverse=True)
print("parameters above 1e-3: %d of %d" % (sum(1 for r in rows if max(r[0], r[1]) > 1e-3), len(rows)))
for r in rows[:int(os.environ.get("TOP", "60"))]:
    print("%.3e %.3e  %-60s %s" % r)
