"""KPConv-FPN point encoder assembled from the HIP kernels
(reference: model/kpconv/kp_backbone.py:79-128, modules.py:115-240, kpconv.py:79-122)."""
import os
from typing import Dict, List

import torch

from . import ops
from .spec import DECODERS, ENCODER, GN_GROUPS, KPBlock

LRELU = 0.1
# Opt-in (COFI_KPCONV_FUSED=1): narrow KPConv layers as ONE kernel (cofi_kpconv_fused).  Correct and 470 MB / frame lighter on the
# fabric, but measured SLOWER on MI355X (490 vs 503 frames/s): 320-1280 eight-wave workgroups that alternate between a gather phase
# and a GEMM phase quantise badly on 256 CUs, while the two-kernel form spreads 20 480 one-wave queries evenly.
FUSED_KPCONV = os.environ.get("COFI_KPCONV_FUSED", "0") == "1"
# The KPConv aggregate (M, 15 C) has ONE reader, the part-2 GEMM.  In stack mode (>= 4 frames per submission) it is written as bf16
# hi / lo planes (same rounding as the GEMM's own split: identical products) and that GEMM runs on gemm_planes_kernel - both operands
# travel global -> LDS by LDS-DMA, no conversion work, no register staging (csrc/gemm_planes.inc): 5-28 % faster per launch on MI355X,
# +2.5 % frames/s at batch 16.  A single frame keeps the fp32 aggregate: measured with four frames in flight the planes path is
# 577-582 vs 583-584 frames/s (its 64-128 KB of LDS per workgroup keep other frames' kernels off the CU, and the aggregation's 8-byte
# plane stores cost what the GEMMs win).  COFI_KPCONV_AGG_PLANES=0 / =all forces never / always (A/B runs, tests).
AGG_PLANES = os.environ.get("COFI_KPCONV_AGG_PLANES", "stack")
AGG_PLANES_MIN_FRAMES = 4


def norm_kind(sd) -> str:
    """Which get_norm() variant (modules.py:51-60) the point encoder's state holds: the key layout differs per kind."""
    if "pc_encoder.encoder1_1.norm.norm.weight" in sd:
        return "gn"
    return "bn" if "pc_encoder.encoder1_1.norm.running_mean" in sd else "ln"


def pack_encoder(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """KPConv weights (15,Cin,Cout) -> (Cout, 15*Cin) so that part 2 of the operator
    (kpconv.py:107-110: sum_k agg[:,k,:] @ W[k]) is ONE K-contiguous GEMM; the rest is used in place.
    'bn' configuration (inference = running statistics): every BatchNorm1d is an affine map per channel behind a Linear / KPConv and
    is folded into that layer's weights and bias here: y * s + t with s = gamma / sqrt(var + eps), t = beta - mean * s."""
    out = {}
    kind = norm_kind(sd)
    fold = {}   # prefix of the layer a BatchNorm follows -> (s, t)
    if kind == "bn":
        for k in sd:
            if k.startswith("pc_encoder.") and k.endswith("running_mean"):
                np_ = k[:-len("running_mean")]                      # "...unary1.norm." / "...norm_conv." / "...encoder1_1.norm."
                s_ = sd[np_ + "weight"].double() / torch.sqrt(sd[np_ + "running_var"].double() + 1e-5)
                t_ = sd[np_ + "bias"].double() - sd[k].double() * s_
                if np_.endswith("norm_conv."):
                    layer = np_[:-len("norm_conv.")] + "KPConv."
                elif (np_[:-len("norm.")] + "KPConv.weights") in sd:   # ConvBlock: X.KPConv + X.norm
                    layer = np_[:-len("norm.")] + "KPConv."
                else:
                    layer = np_[:-len("norm.")] + "mlp."            # UnaryBlock: X.mlp + X.norm
                fold[layer] = (s_, t_)
    for k, v in sd.items():
        if not k.startswith("pc_encoder."):
            continue
        layer = k.rsplit(".", 1)[0] + "."
        st = fold.get(layer)
        if k.endswith("KPConv.weights"):
            w = v.permute(2, 0, 1).reshape(v.shape[2], -1)
            out[k] = (w.double() * st[0][:, None]).float().contiguous() if st else w.contiguous()
        elif st is not None and k.endswith("mlp.weight"):
            out[k] = (v.double() * st[0][:, None]).float().contiguous()
        elif st is not None and k.endswith(("mlp.bias", "KPConv.bias")):
            out[k] = (v.double() * st[0] + st[1]).float().contiguous()
        else:
            out[k] = v.contiguous()
    out["pc_encoder.__norm__"] = kind
    return out


def _gn_width(C: int) -> int:
    """Statistics table width for a GroupNorm(32, C) output: one entry per group and slab (256 bytes per slab) when the group
    width is a power of two the GEMM epilogue can fold (<= 64 columns), else per column."""
    cpg = C // GN_GROUPS
    return cpg if (C % GN_GROUPS == 0 and 1 <= cpg <= 64 and (cpg & (cpg - 1)) == 0) else 1


def _stats(y, part, frames: int, width: int = 1):
    """GroupNorm statistics per frame, still as the GEMM's statistics partials (stack mode: slabs must not straddle frames)."""
    if frames > 1 and (part.shape[0] % frames or (y.shape[0] // frames) % 64):
        raise ops._lib.CofiError("stack mode needs per-frame row counts that are multiples of the 64-row statistics slab (%d rows, %d frames)"
                                 % (y.shape[0], frames))
    return ops.ColStats(part, y.shape[0], GN_GROUPS, frames, width=width)


def _kpconv(P, p: str, feats, q_pts, s_pts, idx, sigma: float, frames: int = 1, order=None):
    """-> (KPConv output, its GroupNorm statistics).  The statistics come out of the producing kernel's epilogue (partials per
    slab and group) instead of another pass over the activation.  Aggregate + GEMM; with COFI_KPCONV_FUSED=1 the narrow layers
    (32 / 64 channels in = out) run as ONE kernel and the (M, 15 C) aggregate never reaches memory (slower, see FUSED_KPCONV)."""
    w = P[p + "KPConv.weights"]
    sw = _gn_width(w.shape[0])
    C = feats.shape[1]
    if FUSED_KPCONV and isinstance(w, ops.SplitW) and w.shape[0] == C and ops.kpconv_fused_slab_rows(C, idx.shape[0] // frames, frames):
        y, part, sr = ops.kpconv_fused(feats, q_pts, s_pts, idx, P[p + "KPConv.kernel_points"], sigma, w, P[p + "KPConv.bias"], stat_width=sw,
                                       frames=frames, order=order)
        return y, ops.ColStats(part, y.shape[0], GN_GROUPS, frames, width=sw, slab_rows=sr)
    planes = (ops.gemm_mode() == "bf16x3" and isinstance(w, ops.SplitW)
              and (AGG_PLANES == "all" or (AGG_PLANES not in ("0", "off") and frames >= AGG_PLANES_MIN_FRAMES)))
    agg, cnt = ops.kpconv_aggregate(feats, q_pts, s_pts, idx, P[p + "KPConv.kernel_points"], sigma, frames=frames, order=order, planes=planes)
    y, part = ops.gemm_colstats(agg, w, bias=P[p + "KPConv.bias"], rowdiv=cnt, stat_width=sw)
    return y, _stats(y, part, frames, sw)


def _unary_raw(P, p: str, x, frames: int = 1):
    """Linear of a UnaryBlock (modules.py:76,89) -> (raw output, GroupNorm statistics).  `x` may be an ops.Normed: the
    previous layer's GroupNorm + LeakyReLU is then applied by this GEMM's operand loader."""
    w = P[p + "mlp.weight"]
    sw = _gn_width(w.shape[0])
    y, part = ops.gemm_colstats(x, w, bias=P[p + "mlp.bias"], stat_width=sw, frames=frames)
    return y, _stats(y, part, frames, sw)


def _unary(P, p: str, x, slope: float, out=None, frames: int = 1, want_row_pos: bool = False):
    y, st = _unary_raw(P, p, x, frames)
    return ops.group_norm_apply(y, st, P[p + "norm.norm.weight"], P[p + "norm.norm.bias"], slope=slope, out=out, frames=frames,
                                want_row_pos=want_row_pos)


def run_block(P, blk: KPBlock, feats, q_pts, s_pts, idx, out=None, concurrent: bool = True, frames: int = 1, order=None):
    p = "pc_encoder.%s." % blk.name
    if blk.kind == "conv":  # modules.py:155-159
        y, st = _kpconv(P, p, feats, q_pts, s_pts, idx, blk.sigma, frames, order)
        return ops.group_norm_apply(y, st, P[p + "norm.norm.weight"], P[p + "norm.norm.bias"], slope=LRELU, out=out, frames=frames)
    # modules.py:222-240.  The shortcut (max-pool / Linear+GN statistics) only meets the main branch in the
    # final fused normalise+add+LeakyReLU: it runs on a side stream.
    has_branch = blk.strided or blk.has_shortcut_unary
    with ops.Branch(feats.device, 1, enabled=concurrent and has_branch) as br:
        sc = ops.neighbor_maxpool(feats, idx, frames=frames, order=order) if blk.strided else feats
        ys = sts = None
        if blk.has_shortcut_unary:
            ys, sts = _unary_raw(P, p + "unary_shortcut.", sc, frames)
    # unary1 feeds the KPConv gather: its normalisation is materialised, and the apply kernel also emits the per-row flag
    # the aggregation needs
    x = _unary(P, p + "unary1.", feats, LRELU, frames=frames, want_row_pos=True) if blk.cin != blk.mid else feats
    y, st = _kpconv(P, p, x, q_pts, s_pts, idx, blk.sigma, frames, order)
    # norm_conv + LeakyReLU (modules.py:232-233) is consumed by unary2's Linear only: applied by that GEMM's operand loader
    x = ops.Normed(y, st, P[p + "norm_conv.norm.weight"], P[p + "norm_conv.norm.bias"], LRELU)
    y2, st2 = _unary_raw(P, p + "unary2.", x, frames)
    br.join(sc, ys, None if sts is None else sts.part)
    g2, b2 = P[p + "unary2.norm.norm.weight"], P[p + "unary2.norm.norm.bias"]
    if blk.has_shortcut_unary:
        return ops.group_norm_apply(y2, st2, g2, b2, slope=LRELU, res=ys, res_stats=sts,
                                    res_gamma=P[p + "unary_shortcut.norm.norm.weight"],
                                    res_beta=P[p + "unary_shortcut.norm.norm.bias"], out=out, frames=frames)
    return ops.group_norm_apply(y2, st2, g2, b2, slope=LRELU, res=sc, out=out, frames=frames)


# ---- 'bn' / 'ln' configurations of get_norm() (modules.py:51-60).  Row-wise normalisations: nothing to accumulate across rows, so these
# paths are plain kernel sequences - BatchNorm (running statistics) is already folded into the weights, the LeakyReLU and the residual
# join ride in the GEMM epilogue; LayerNorm is the row kernel with LeakyReLU / residual fused.
def _ln(P, np_: str, y, slope: float, res=None, out=None):
    return ops.layer_norm_act(y, P[np_ + "weight"], P[np_ + "bias"], slope=slope, res=res, res_first=True, out=out)


def _unary_plain(P, kind: str, p: str, x, slope: float, res=None, out=None):
    """UnaryBlock (modules.py:63-94): Linear -> norm -> LeakyReLU(slope) with an optional residual joined before the activation."""
    w, b = P[p + "mlp.weight"], P[p + "mlp.bias"]
    if kind == "bn":
        act = ops.ACT_LEAKY01 if slope != 1.0 else ops.ACT_NONE
        if res is None:
            return ops.gemm(x, w, bias=b, act=act, out=out)
        return ops.conv2d_nhwc(x, x.shape[0], 1, w, 1, stride=1, pad=0, bias=b, res=res, act=act, out=out)[0]   # 1x1 "convolution" = GEMM + residual
    return _ln(P, p + "norm.", ops.gemm(x, w, bias=b), slope, res=res, out=out)


def _kpconv_plain(P, kind: str, p: str, np_: str, feats, q_pts, s_pts, idx, sigma, frames, order, out=None):
    agg, cnt = ops.kpconv_aggregate(feats, q_pts, s_pts, idx, P[p + "KPConv.kernel_points"], sigma, frames=frames, order=order)
    w, b = P[p + "KPConv.weights"], P[p + "KPConv.bias"]
    if kind == "bn":
        return ops.gemm(agg, w, bias=b, rowdiv=cnt, act=ops.ACT_LEAKY01, out=out)
    return _ln(P, np_, ops.gemm(agg, w, bias=b, rowdiv=cnt), LRELU, out=out)


def run_block_plain(P, kind: str, blk: KPBlock, feats, q_pts, s_pts, idx, out=None, frames: int = 1, order=None):
    p = "pc_encoder.%s." % blk.name
    if blk.kind == "conv":   # modules.py:155-159
        return _kpconv_plain(P, kind, p, p + "norm.", feats, q_pts, s_pts, idx, blk.sigma, frames, order, out=out)
    x = _unary_plain(P, kind, p + "unary1.", feats, LRELU) if blk.cin != blk.mid else feats   # modules.py:222-240
    x = _kpconv_plain(P, kind, p, p + "norm_conv.", x, q_pts, s_pts, idx, blk.sigma, frames, order)
    sc = ops.neighbor_maxpool(feats, idx, frames=frames, order=order) if blk.strided else feats
    if blk.has_shortcut_unary:
        sc = _unary_plain(P, kind, p + "unary_shortcut.", sc, 1.0)
    return _unary_plain(P, kind, p + "unary2.", x, LRELU, res=sc, out=out)


def run_fpn(P, points: List[torch.Tensor], neighbors, subsampling, upsampling, feats, taps=None, frames: int = 1, order=None, l2norm_fine: bool = False):
    """`order` (optional): per stage, the frame-local processing order of the stage's points (spatially sorted)."""
    """Returns [latent_s2 (N1,64), latent_s3 (N2,512), latent_s4 (N3,1024), feats_s5 (N4,2048)].
    The last block of stages 1..3 writes directly into the right part of the decoder's concat
    buffer (kp_backbone.py:112,117,122 torch.cat)."""
    dev = feats.device
    dec_in = {name: cin for name, cin, _, _ in DECODERS}
    # concat buffers: [upsampled deeper latent | stage features]
    cat = {3: torch.empty((points[3].shape[0], dec_in["decoder4"]), dtype=torch.float32, device=dev),
           2: torch.empty((points[2].shape[0], dec_in["decoder3"]), dtype=torch.float32, device=dev),
           1: torch.empty((points[1].shape[0], dec_in["decoder2"]), dtype=torch.float32, device=dev)}
    stage_width = {1: 256, 2: 512, 3: 1024}
    last_of_stage = {}
    for blk in ENCODER:
        last_of_stage[blk.stage] = blk.name
    x = feats
    stage_out = {}
    kind = P.get("pc_encoder.__norm__", "gn")
    for blk in ENCODER:
        st = blk.stage
        if blk.strided:
            q, s, idx = points[st], points[st - 1], subsampling[st - 1]
        else:
            q, s, idx = points[st], points[st], neighbors[st]
        out = None
        if st in cat and last_of_stage[st] == blk.name:
            w = stage_width[st]
            out = cat[st][:, cat[st].shape[1] - w:]
        if kind == "gn":
            x = run_block(P, blk, x, q, s, idx, out=out, frames=frames, order=None if order is None else order[st])
        else:
            x = run_block_plain(P, kind, blk, x, q, s, idx, out=out, frames=frames, order=None if order is None else order[st])
        stage_out[st] = x
        if taps is not None:
            taps[blk.name] = x
    s5 = stage_out[4]
    ops.gather_rows(s5, upsampling[3], out=cat[3][:, :2048], frames=frames)
    dec = (lambda name, x_: _unary(P, name, x_, LRELU, frames=frames)) if kind == "gn" else (lambda name, x_: _unary_plain(P, kind, name, x_, LRELU))
    l4 = dec("pc_encoder.decoder4.", cat[3])
    ops.gather_rows(l4, upsampling[2], out=cat[2][:, :1024], frames=frames)
    l3 = dec("pc_encoder.decoder3.", cat[2])
    ops.gather_rows(l3, upsampling[1], out=cat[1][:, :512], frames=frames)
    # l2norm_fine: the stage-2 latent leaves L2-normalised (network.py:83, its only reader) from this GEMM's epilogue
    l2 = ops.gemm(cat[1], P["pc_encoder.decoder2.mlp.weight"], bias=P["pc_encoder.decoder2.mlp.bias"], l2norm=l2norm_fine)
    if taps is not None:
        taps.update(decoder4=l4, decoder3=l3, decoder2=l2)
    return [l2, l3, l4, s5]
