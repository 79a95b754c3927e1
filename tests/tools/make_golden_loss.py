"""Generate tests/golden/loss_ref.npz by RUNNING THE REFERENCE'S model/loss.py (development container only):

    python tests/tools/make_golden_loss.py

Seeded inputs shaped like train.py:233-282 (unit-norm descriptor columns, a correspondence mask with a few positives per row,
sigmoid scores, 4x4 patches), the three loss values, the `dists` matrix desc_loss returns, and - through torch.autograd on the
reference's own expressions - the gradients of each loss w.r.t. its tensor inputs.  The fixture is data: inputs and expected outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, os.environ.get("COFI_REFERENCE_ROOT", "/root/reference"))

from model.loss import desc_loss, fine_circle_loss, overlap_loss  # noqa: E402  (the reference's own functions)


def unit_cols(g, C, K):
    x = torch.from_numpy(g.standard_normal((C, K)).astype(np.float32))
    return x / x.norm(dim=0, keepdim=True)


def main():
    out = {}
    for tag, (K, C, Cf) in {"kitti": (64, 128, 64), "nuscenes": (32, 128, 64), "odd": (37, 20, 12)}.items():
        g = np.random.default_rng(7 + K)
        img, pc = unit_cols(g, C, K).requires_grad_(), unit_cols(g, C, K).requires_grad_()
        mask = torch.zeros(K, K)
        for i in range(K):   # the diagonal pair and a few neighbours correspond (train.py:251); two rows have no positive at all
            if i % 17 != 5:
                mask[i, i] = 1
                if i + 1 < K and i % 3 == 0:
                    mask[i, i + 1] = 1
        loss, dists = desc_loss("cpu", img, pc, mask, pos_margin=0.2, neg_margin=1.8)   # options.py:42-43
        loss.backward()
        out.update({tag + "_img": img.detach(), tag + "_pc": pc.detach(), tag + "_mask": mask, tag + "_desc_loss": loss.detach(),
                    tag + "_dists": dists.detach(), tag + "_desc_gimg": img.grad, tag + "_desc_gpc": pc.grad})
        s_in = torch.sigmoid(torch.from_numpy(g.standard_normal(K).astype(np.float32)) * 2 + 1).requires_grad_()
        s_out = torch.sigmoid(torch.from_numpy(g.standard_normal(K).astype(np.float32)) * 2 - 1).requires_grad_()
        lo = overlap_loss("cpu", s_in, s_out)
        lo.backward()
        out.update({tag + "_sin": s_in.detach(), tag + "_sout": s_out.detach(), tag + "_overlap_loss": lo.detach(), tag + "_overlap_gin": s_in.grad,
                    tag + "_overlap_gout": s_out.grad})
        patches = torch.from_numpy(g.standard_normal((K, Cf, 4, 4)).astype(np.float32))
        patches = (patches / patches.norm(dim=1, keepdim=True)).requires_grad_()
        fpc = torch.from_numpy(g.standard_normal((K, Cf)).astype(np.float32))
        rel = torch.from_numpy(g.integers(0, 16, K))
        with torch.no_grad():   # make the true pixel resemble the point descriptor, as a trained network would
            pv = patches.detach().reshape(K, Cf, 16)
            fpc = fpc + 2.0 * pv[torch.arange(K), :, rel]
            fpc = fpc / fpc.norm(dim=1, keepdim=True)
        fpc.requires_grad_()
        lf = fine_circle_loss("cpu", patches, fpc, rel, num_kpt=K)
        lf.backward()
        out.update({tag + "_patches": patches.detach(), tag + "_fpc": fpc.detach(), tag + "_rel": rel, tag + "_fine_loss": lf.detach(),
                    tag + "_fine_gpatches": patches.grad, tag + "_fine_gpc": fpc.grad})
    np.savez_compressed(os.path.join(GOLD, "loss_ref.npz"), **{k: v.numpy() for k, v in out.items()})
    print("loss_ref.npz: %d arrays; desc %.6f overlap %.6f fine %.6f (kitti)" % (len(out), float(out["kitti_desc_loss"]), float(out["kitti_overlap_loss"]),
                                                                                  float(out["kitti_fine_loss"])))


if __name__ == "__main__":
    main()
