"""Diagnostic (GPU box): gradients of the train step at intermediate tensors - HIP path against torch.autograd through the CPU oracle (fp32)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cofi_oracle as O  # noqa: E402
import loss_oracle as LO  # noqa: E402
from common import frame_inputs, load_golden  # noqa: E402

from cofii2p_amd import train_forward as TF  # noqa: E402
from cofii2p_amd.network import CoFiI2P  # noqa: E402
from cofii2p_amd.spec import synth_state_dict  # noqa: E402
from cofii2p_amd.train_step import step_losses  # noqa: E402

T = torch.from_numpy


def keep(store, name, t):
    if t.requires_grad:
        t.retain_grad()
    store[name] = t
    return t


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "bn"
    gold = load_golden("train_ref.npz" if kind == "gn" else "train_ref_%s.npz" % kind)
    fr, data = frame_inputs(int(gold["frame_id"]), int(gold["num_points"]), int(gold["pyr_seed"]))
    lab = {k[4:]: T(gold[k]) for k in gold.files if k.startswith("lab_")}
    # ---------------- oracle, fp32 autograd
    sd = {k: T(v).clone() for k, v in synth_state_dict(norm=kind).items()}
    for n in [str(n) for n in gold["g_names"]]:
        sd[n] = sd[n].clone().requires_grad_()
    so = {}
    o_tr, o_mlp, o_sh = O.transformer, O.pc_feature_mlp, O.score_head

    def tr(sd_, a, b, **kw):
        a, b = keep(so, "tr_in_img", a), keep(so, "tr_in_pc", b)
        x, y = o_tr(sd_, a, b, **kw)
        return keep(so, "tr_out_img", x), keep(so, "tr_out_pc", y)

    def mlp(sd_, x):
        return keep(so, "mlp_out", o_mlp(sd_, keep(so, "s5", x)))

    def sh(sd_, p, tok):
        return keep(so, "score_" + p[:2], o_sh(sd_, p, tok))

    O.transformer, O.pc_feature_mlp, O.score_head = tr, mlp, sh
    img = T(fr.img)[None]
    outs = O.forward(sd, data, img, lab["fine_center_kpt_coors"].float(), lab["fine_pc_inline_index"], "train", train_bn=True)
    O.transformer, O.pc_feature_mlp, O.score_head = o_tr, o_mlp, o_sh
    img_f, pc_f, _s, pc_s, patches, fine_pc = outs[:6]
    kp, ko, ci = lab["pc_kpt_idx"], lab["pc_outline_idx"], lab["coarse_img_kpt_idx"]
    l_desc, _ = LO.desc_loss(img_f.reshape(img_f.shape[1], -1)[:, ci], pc_f[:, kp], T(gold["mask"]), float(gold["pos_margin"]), float(gold["neg_margin"]))
    l_coarse = LO.overlap_loss(pc_s[0, 0, kp], pc_s[0, 0, ko])
    rel = lab["fine_xy"] - lab["fine_center_kpt_coors"] + 2
    l_fine = LO.fine_circle_loss(patches, fine_pc, rel[1] * 4 + rel[0])
    keep(so, "pc_desc", pc_f), keep(so, "img_desc", img_f), keep(so, "pc_score", pc_s)
    (l_desc + l_coarse + l_fine).backward()
    # ---------------- HIP
    sh_ = {}
    h_tr, h_mlp, h_sh = TF.transformer, TF.pc_feature_mlp, TF.score_head

    def htr(P, a, b):
        a, b = keep(sh_, "tr_in_img", a), keep(sh_, "tr_in_pc", b)
        x, y = h_tr(P, a, b)
        return keep(sh_, "tr_out_img", x), keep(sh_, "tr_out_pc", y)

    def hmlp(P, x):
        return keep(sh_, "mlp_out", h_mlp(P, keep(sh_, "s5", x)))

    def hsh(P, p, tok):
        return keep(sh_, "score_" + p[:2], h_sh(P, p, tok))

    TF.transformer, TF.pc_feature_mlp, TF.score_head = htr, hmlp, hsh
    dd = {k: [t.cuda() for t in v] for k, v in data.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
    dd["feats"] = data["feats"].cuda()

    class Opt:
        img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, kind

    class SOpt:
        dist_thres, pos_margin, neg_margin = float(gold["dist_thres"]), float(gold["pos_margin"]), float(gold["neg_margin"])

    m = CoFiI2P(Opt(), arithmetic="bf16x6").cuda()
    m.train()
    houts, _, losses = step_losses(m, dd, img.cuda(), {k: v.cuda() for k, v in lab.items()}, SOpt)
    keep(sh_, "pc_desc", houts[1]), keep(sh_, "img_desc", houts[0]), keep(sh_, "pc_score", houts[3])
    sum(losses).backward()
    print("losses oracle", float(l_desc), float(l_coarse), float(l_fine), " HIP", [float(v) for v in losses])
    for name in so:
        a, b = so[name], sh_[name].detach().cpu()
        b = b.reshape(a.shape) if b.numel() == a.numel() and b.shape != a.shape and name.startswith(("score", "pc_score")) else b
        if b.shape != a.shape:
            b = b.t() if b.t().shape == a.shape else b.reshape(a.shape)
        fv = float((a.detach() - b).norm() / a.detach().norm())
        ga, gb = so[name].grad, sh_[name].grad
        if ga is None or gb is None:
            print("%-12s value %.2e   grad: %s / %s" % (name, fv, ga is not None, gb is not None))
            continue
        gb = gb.cpu()
        if gb.shape != ga.shape:
            gb = gb.t() if gb.t().shape == ga.shape else gb.reshape(ga.shape)
        d = (ga - gb)
        rows = d.reshape(d.shape[0], -1).norm(dim=1) if d.dim() > 1 else d.abs()
        print("%-12s value %.2e   grad %.2e   (|g| %.3e; worst row %d: %.2e of |g|)" % (name, fv, float(d.norm() / ga.norm()), float(ga.norm()), int(rows.argmax()), float(rows.max() / ga.norm())))


if __name__ == "__main__":
    main()


def score_hidden(kind="bn"):
    """pre-ReLU values of the point score head (fp64, oracle tokens): the smallest |value| in rows that carry a loss gradient"""
    gold = load_golden("train_ref.npz" if kind == "gn" else "train_ref_%s.npz" % kind)
    fr, data = frame_inputs(int(gold["frame_id"]), int(gold["num_points"]), int(gold["pyr_seed"]))
    lab = {k[4:]: T(gold[k]) for k in gold.files if k.startswith("lab_")}
    sd = {k: T(v).clone() for k, v in synth_state_dict(norm=kind).items()}
    taps = {}
    o_sh = O.score_head

    def sh(sd_, p, tok):
        taps[p] = tok
        return o_sh(sd_, p, tok)

    O.score_head = sh
    with torch.no_grad():
        O.forward(sd, data, T(fr.img)[None], lab["fine_center_kpt_coors"].float(), lab["fine_pc_inline_index"], "train", train_bn=True)
    O.score_head = o_sh
    tok = taps["pc_score_layer."].double()
    rows = torch.cat([lab["pc_kpt_idx"], lab["pc_outline_idx"]]).unique()

    def inorm(y):
        var, mean = torch.var_mean(y, dim=0, unbiased=False, keepdim=True)
        return (y - mean) * torch.rsqrt(var + 1e-5)

    p = "pc_score_layer."
    x1 = inorm(tok @ sd[p + "0.weight"].reshape(128, 128).double().t())
    x2 = inorm(torch.relu(x1) @ sd[p + "3.weight"].reshape(64, 128).double().t())
    for name, x in (("layer 0", x1), ("layer 3", x2)):
        a = x.abs()
        i = int(a.reshape(-1).argmin())
        print("%s: smallest |pre-ReLU| overall %.3e at row %d; in gradient rows %.3e at row %d; row 100 min %.3e" % (
            name, float(a.min()), i // x.shape[1], float(a[rows].min()), int(rows[int(a[rows].min(dim=1).values.argmin())]), float(a[100].min())))
