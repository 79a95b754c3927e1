"""Image branch: ResNet-34 (affine-less InstanceNorm) + two bilinear up-sample / ResidualConv stages
(reference: model/imagenet.py:119-217, 377-444).

Feature maps are NHWC = pixel-major (H*W, C) matrices like every other activation of the network; a convolution is an
implicit GEMM on the MFMA kernel (`cofi_conv2d_nhwc_fused`), whose epilogue emits the per-channel statistics partials
InstanceNorm needs (or fuses the folded-BatchNorm bias, ReLU and the skip convolution of a ResidualConv).  The
InstanceNorm + ReLU between the two convolutions of a BasicBlock is applied by the second convolution's operand loader
(ops.Normed); InstanceNorm + ReLU + residual at the end of a block is the GroupNorm apply kernel with one group per
channel.  No vendor library, no layout transposes, bit-reproducible.
"""
from typing import Dict, List

import torch

from . import ops
from .spec import RESNET_LAYERS


def _nhwc_weight(w: torch.Tensor, kpad: int = 0) -> torch.Tensor:
    """OIHW -> (O, kh*kw*I) with k = (dy*kw + dx)*I + c (the implicit-GEMM column order), optionally zero padded."""
    o = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    if kpad and kpad > o.shape[1]:
        o = torch.cat([o, torch.zeros((o.shape[0], kpad - o.shape[1]), dtype=o.dtype, device=o.device)], 1)
    return o.contiguous()


def pack_image(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if k.startswith("img_encoder.backbone.") and k.endswith("weight") and v.dim() == 4:
            out[k + ".nhwc"] = _nhwc_weight(v, 160 if k.endswith("backbone.conv1.weight") else 0)
    eps = 1e-5
    for name in ("img_upsample_1", "img_upsample_2"):
        for j in (0, 1):
            p = "%s.conv.%d." % (name, j)
            for conv, bn in (("conv1", "bn1"), ("conv2", "bn2"), ("conv_skip.0", "conv_skip.1")):
                w = sd[p + conv + ".weight"]
                scale = sd[p + bn + ".weight"] * torch.rsqrt(sd[p + bn + ".running_var"] + eps)
                out[p + conv + ".w.nhwc"] = _nhwc_weight(w * scale[:, None, None, None])
                out[p + conv + ".b"] = (sd[p + bn + ".bias"] - sd[p + bn + ".running_mean"] * scale).contiguous()
            # conv_skip and conv1 read the same input (imagenet.py:399-402): their filters run as ONE implicit GEMM [skip | conv1]
            out[p + "skip_conv1.w.nhwc"] = torch.cat([out[p + "conv_skip.0.w.nhwc"], out[p + "conv1.w.nhwc"]], 0).contiguous()
            out[p + "skip_conv1.b"] = torch.cat([out[p + "conv_skip.0.b"], out[p + "conv1.b"]], 0).contiguous()
            del out[p + "conv_skip.0.w.nhwc"], out[p + "conv1.w.nhwc"]
    return out


def _slabs_ok(y, part, frames):
    """the 64-row statistics slabs of the producing convolution do not straddle frames"""
    return frames == 1 or (part.shape[0] % frames == 0 and (y.shape[0] // frames) % 64 == 0)


def _in_relu(y, part, res=None, res_part=None, frames: int = 1):
    """relu(InstanceNorm(y) + [res | InstanceNorm(res)]) on (P, C) maps: one group per channel (per frame)."""
    P, C = y.shape
    if not _slabs_ok(y, part, frames) or (res_part is not None and not _slabs_ok(res, res_part, frames)):
        # tiny maps whose pixel count is not a multiple of the statistics slab (5x16 map of layer4): per-frame statistics by a
        # separate pass over the map (still one launch for all frames)
        st = ops.group_stats(y, C, frames=frames)
        rst = None if res_part is None else ops.group_stats(res, C, frames=frames)
        return ops.group_norm_apply(y, st, slope=0.0, res=res, res_stats=rst, frames=frames)
    st = ops.ColStats(part, P, C, frames)
    rst = None if res_part is None else ops.ColStats(res_part, P, C, frames)
    return ops.group_norm_apply(y, st, slope=0.0, res=res, res_stats=rst, frames=frames)


def _resnet_layer_nhwc(P, li: int, blocks: int, stride: int, x, H: int, W: int, frames: int):
    p = "img_encoder.backbone."
    for b in range(blocks):
        q = "%slayer%d.%d." % (p, li, b)
        st = stride if b == 0 else 1
        y1, part1, Ho, Wo = ops.conv2d_nhwc(x, H, W, P[q + "conv1.weight.nhwc"], 3, st, 1, colstats=True, frames=frames)
        if _slabs_ok(y1, part1, frames):
            # relu(IN(y1)) is read by conv2 only (imagenet.py:60-64): its operand loader applies it
            a = ops.Normed(y1, ops.ColStats(part1, y1.shape[0], y1.shape[1], frames), slope=0.0)
        else:
            a = _in_relu(y1, part1, frames=frames)
        y2, part2, _, _ = ops.conv2d_nhwc(a, Ho, Wo, P[q + "conv2.weight.nhwc"], 3, 1, 1, colstats=True, frames=frames)
        if (q + "downsample.0.weight.nhwc") in P:
            d, partd, _, _ = ops.conv2d_nhwc(x, H, W, P[q + "downsample.0.weight.nhwc"], 1, st, 0, colstats=True, frames=frames)
            x = _in_relu(y2, part2, res=d, res_part=partd, frames=frames)
        else:
            x = _in_relu(y2, part2, res=x, frames=frames)
        H, W = Ho, Wo
    return x, H, W


def resnet34_nhwc(P, img: torch.Tensor, full: bool = True, tail_branch=None):
    """imagenet.py:196-217 on NHWC maps.  img (frames,3,H,W).  Returns ([s2, s4, s8, s16, s32, gap], [(H,W) per map]).
    layer3, layer4 and the average pool feed nothing downstream (network.py:87-89 only names them): when the caller hands
    in a `tail_branch` (ops.Branch) they are enqueued on its side stream and the caller joins it whenever it likes."""
    p = "img_encoder.backbone."
    frames = img.shape[0]
    col, H, W = ops.im2col_stem(img.contiguous())
    y, part = ops.gemm_colstats(col, P[p + "conv1.weight.nhwc"])
    x = _in_relu(y, part, frames=frames)
    outs, dims = [x], [(H, W)]
    x, H, W = ops.maxpool3x3s2_nhwc(x, H, W, frames)
    for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS[:2], start=1):
        x, H, W = _resnet_layer_nhwc(P, li, blocks, stride, x, H, W, frames)
        outs.append(x)
        dims.append((H, W))
    if not full:
        return outs + [None, None, None], dims + [None, None, (1, 1)]

    def tail(x, H, W):
        for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS[2:], start=3):
            x, H, W = _resnet_layer_nhwc(P, li, blocks, stride, x, H, W, frames)
            outs.append(x)
            dims.append((H, W))
        outs.append(ops.col_mean(x, frames))  # AdaptiveAvgPool2d(1): unused downstream (network.py:87)
        dims.append((1, 1))

    if tail_branch is not None:
        with tail_branch:
            tail(x, H, W)
    else:
        tail(x, H, W)
    return outs, dims


def _residual_conv_nhwc(P, p: str, x, H, W, frames: int = 1, l2norm: bool = False):
    """imagenet.py:397-411: the skip convolution and conv1 share their input and run as ONE implicit GEMM (filters stacked
    [skip | conv1], ReLU on the conv1 half only), conv2 reads that half as a strided view; folded-BN bias, ReLU and the skip add
    live in the epilogues."""
    C = P[p + "conv2.b"].shape[0]
    z, _, _ = ops.conv2d_nhwc(x, H, W, P[p + "skip_conv1.w.nhwc"], 3, 1, 1, bias=P[p + "skip_conv1.b"], act=ops.ACT_RELU, act_col0=C, frames=frames)
    out, _, _ = ops.conv2d_nhwc(z[:, C:], H, W, P[p + "conv2.w.nhwc"], 3, 1, 1, bias=P[p + "conv2.b"], res=z[:, :C], act=ops.ACT_RELU, frames=frames,
                                l2norm=l2norm and C <= 128)
    return out if (not l2norm or C <= 128) else ops.l2norm_rows(out)


def upsample_stage_nhwc(P, name: str, low, h, w, skip, frames: int = 1, l2norm: bool = False):
    """imagenet.py:431-444 on NHWC maps: low (h*w, C1), skip (4hw, C2) -> (4hw, Cout).  l2norm: the pixels of the result are
    L2-normalised (network.py:130) by the last convolution's epilogue."""
    x = ops.upsample2x_cat_nhwc(low, h, w, skip, frames)
    x = _residual_conv_nhwc(P, name + ".conv.0.", x, 2 * h, 2 * w, frames)
    return _residual_conv_nhwc(P, name + ".conv.1.", x, 2 * h, 2 * w, frames, l2norm=l2norm)
