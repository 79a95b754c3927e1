"""Diagnostic (GPU box): the point encoder of the differentiable forward per opt.norm against the CPU oracle evaluated in float64,
block by block (relative Frobenius error), next to the oracle's own fp32 error.  python tools/diag_norm_train.py [gn bn ln]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cofi_oracle as O  # noqa: E402
from common import frame_inputs  # noqa: E402

from cofii2p_amd import autograd as ag, ops, train_forward as TF  # noqa: E402
from cofii2p_amd.network import CoFiI2P  # noqa: E402
from cofii2p_amd.spec import synth_state_dict  # noqa: E402


def main():
    kinds = sys.argv[1:] or ["gn", "bn", "ln"]
    fr, data = frame_inputs(1, 2048, 11)
    for kind in kinds:
        sd = {k: torch.from_numpy(v).clone() for k, v in synth_state_dict(norm=kind).items()}
        O.BN_TRAIN = True
        t32, t64 = {}, {}
        O.kpconv_fpn({k: v.clone() for k, v in sd.items()}, data, taps=t32)
        dbl = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
        O.kpconv_fpn({k: dbl(v.clone()) for k, v in sd.items()}, {k: ([dbl(t) for t in v] if isinstance(v, list) else dbl(v)) for k, v in data.items()}, taps=t64)
        O.BN_TRAIN = False

        class Opt:
            img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, kind

        for arith in ("bf16x6", "f32"):
            m = CoFiI2P(Opt(), arithmetic=arith).cuda()
            m.train()
            P, B = dict(m.named_parameters()), dict(m.named_buffers())
            dd = {k: [t.cuda() for t in v] for k, v in data.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
            taps = {}
            with torch.no_grad(), ops.arithmetic(arith):
                TF.kpconv_fpn(P, B, dd["points"], [m._as_idx32(t) for t in dd["neighbors"]], [m._as_idx32(t) for t in dd["subsampling"]],
                              [m._as_idx32(t) for t in dd["upsampling"]], data["feats"].cuda(), ag.TableCache(), kind, True, taps)
            print("norm %s  arithmetic %s" % (kind, arith))
            for name in t64:
                ref = t64[name]
                e_hip = float((taps[name].double().cpu() - ref).norm() / ref.norm())
                e_o32 = float((t32[name].double() - ref).norm() / ref.norm())
                print("  %-12s rows %5d  C %4d   HIP %.2e   oracle-fp32 %.2e" % (name, ref.shape[0], ref.shape[1], e_hip, e_o32))


if __name__ == "__main__":
    main()
