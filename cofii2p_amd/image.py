"""Image branch: ResNet-34 (affine-less InstanceNorm) + two bilinear up-sample / ResidualConv stages
(reference: model/imagenet.py:119-217, 377-444).

Round-1 status: the dense 3x3/7x7 convolutions, InstanceNorm, max-pool and
bilinear resize run through PyTorch-ROCm's MIOpen/ATen device ops; BatchNorm (eval mode, running
statistics) is folded into the convolution weights once at pack time.  Everything downstream of
the feature maps (L2 normalisation, token layout, matching) is hand-written HIP.
"""
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import ops
from .spec import RESNET_LAYERS


def pack_image(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if k.startswith("img_encoder.backbone.") and k.endswith("weight") and v.dim() == 4:
            out[k] = v.contiguous()
    eps = 1e-5
    for name in ("img_upsample_1", "img_upsample_2"):
        for j in (0, 1):
            p = "%s.conv.%d." % (name, j)
            for conv, bn in (("conv1", "bn1"), ("conv2", "bn2"), ("conv_skip.0", "conv_skip.1")):
                w = sd[p + conv + ".weight"]
                scale = sd[p + bn + ".weight"] * torch.rsqrt(sd[p + bn + ".running_var"] + eps)
                out[p + conv + ".w"] = (w * scale[:, None, None, None]).contiguous()
                out[p + conv + ".b"] = (sd[p + bn + ".bias"] - sd[p + bn + ".running_mean"] * scale).contiguous()
    return out


def resnet34(P, img: torch.Tensor, full: bool = True) -> List[torch.Tensor]:
    """imagenet.py:196-217.  Returns [s2, s4, s8, s16, s32, gap]; with full=False the maps nothing
    downstream reads (layer3, layer4, avg-pool: network.py:87-89) are skipped and returned as None.
    Convolutions / max-pool: MIOpen; InstanceNorm + ReLU + residual tails: one HIP kernel each."""
    p = "img_encoder.backbone."
    x = ops.instance_norm_nchw(F.conv2d(img, P[p + "conv1.weight"], stride=2, padding=3), relu=True)
    outs = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS, start=1):
        if not full and li > 2:
            outs.append(None)
            continue
        for b in range(blocks):
            q = "%slayer%d.%d." % (p, li, b)
            st = stride if b == 0 else 1
            y = ops.instance_norm_nchw(F.conv2d(x, P[q + "conv1.weight"], stride=st, padding=1), relu=True)
            y = F.conv2d(y, P[q + "conv2.weight"], padding=1)
            if (q + "downsample.0.weight") in P:  # relu(IN(y) + IN(downsample(x)))
                x = ops.instance_norm_nchw(y, relu=True, res=F.conv2d(x, P[q + "downsample.0.weight"], stride=st), res_norm=True)
            else:  # relu(IN(y) + x)
                x = ops.instance_norm_nchw(y, relu=True, res=x)
        outs.append(x)
    outs.append(F.adaptive_avg_pool2d(x, 1) if full else None)
    return outs


def _residual_conv(P, p: str, x):
    """imagenet.py:397-411 with the eval-mode BatchNorms folded into the convolutions (pack_image)."""
    skip = F.conv2d(x, P[p + "conv_skip.0.w"], padding=1)
    y = ops.bias_act_nchw(F.conv2d(x, P[p + "conv1.w"], padding=1), P[p + "conv1.b"], relu=True)
    y = F.conv2d(y, P[p + "conv2.w"], padding=1)
    return ops.bias_act_nchw(y, P[p + "conv2.b"], res=skip, res_bias=P[p + "conv_skip.0.b"], relu=True)


def upsample_stage(P, name: str, low, skip):
    """imagenet.py:431-444."""
    x = ops.upsample2x_cat(low, skip)
    return _residual_conv(P, name + ".conv.1.", _residual_conv(P, name + ".conv.0.", x))
