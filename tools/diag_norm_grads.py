"""Diagnostic (GPU box): per parameter group, the error of the HIP train step's gradients against the reference's float64 gradients
(tests/golden/train_ref*.npz), next to the reference's own fp32 deviation.  python tools/diag_norm_grads.py [gn bn ln] [--arith f32]"""
import itertools
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import frame_inputs, load_golden  # noqa: E402

from cofii2p_amd.network import CoFiI2P  # noqa: E402
from cofii2p_amd.train_step import step_losses  # noqa: E402


def main():
    args = sys.argv[1:]
    arith = "bf16x6"
    if "--arith" in args:
        arith = args[args.index("--arith") + 1]
        args = [a for a in args if a not in ("--arith", arith)]
    for kind in args or ["gn", "bn", "ln"]:
        gold = load_golden("train_ref.npz" if kind == "gn" else "train_ref_%s.npz" % kind)
        fr, data = frame_inputs(int(gold["frame_id"]), int(gold["num_points"]), int(gold["pyr_seed"]))
        dd = {k: [t.cuda() for t in v] for k, v in data.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
        dd["feats"] = data["feats"].cuda()
        img = torch.from_numpy(fr.img)[None].cuda()
        batch = {k[4:]: torch.from_numpy(gold[k]).cuda() for k in gold.files if k.startswith("lab_")}

        class Opt:
            img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, kind

        class SOpt:
            dist_thres, pos_margin, neg_margin = float(gold["dist_thres"]), float(gold["pos_margin"]), float(gold["neg_margin"])

        m = CoFiI2P(Opt(), arithmetic=arith).cuda()
        m.train()
        outs, mask, losses = step_losses(m, dd, img, batch, SOpt)
        for n_, t in zip(("img_desc", "pc_desc", "img_score", "pc_score", "patches", "fine_pc"), outs[:6]):
            print("  %s %-9s max abs err %.2e" % (kind, n_, float((t.detach().cpu() - torch.from_numpy(gold["train_" + n_])).abs().max())))
        sum(losses).backward()
        names = [str(n) for n in gold["g_names"]]
        params = dict(m.named_parameters())
        total = float(np.sqrt((gold["g_norm64"] ** 2).sum()))
        rows = []
        for i, name in enumerate(names):
            if not int(gold["g_has"][i]) or float(gold["g_norm64"][i]) < 1e-6 * total:
                continue
            flat = params[name].grad.detach().double().reshape(-1).cpu()
            got, ref = flat[torch.from_numpy(gold["g_pos"][i])].numpy(), gold["g_val64"][i]
            scale = max(np.linalg.norm(ref), float(gold["g_norm64"][i]) * math.sqrt(len(ref) / flat.numel()))
            rows.append((name, float(np.linalg.norm(got - ref) / scale), float(gold["g_err32"][i])))
        key = lambda r: ".".join(r[0].split(".")[:3]) if r[0].startswith("transformer") else r[0].split(".")[0]
        print("norm %s arithmetic %s: group, parameters, median / max error of the HIP gradient, median / max of the reference's fp32 gradient" % (kind, arith))
        for k, grp in itertools.groupby(rows, key):
            grp = list(grp)
            e, r = np.array([g[1] for g in grp]), np.array([g[2] for g in grp])
            print("  %-28s %3d   HIP %.2e / %.2e   ref32 %.2e / %.2e" % (k, len(grp), np.median(e), e.max(), np.median(r), r.max()))


if __name__ == "__main__":
    main()
