"""CoFiI2P.forward(mode='train') with autograd enabled: the differentiable form of the forward.

The same network as `network.CoFiI2P._run_device`, expressed as a torch.autograd graph over the module's own nn.Parameters so that
train.py:224-288 (forward -> losses -> loss.backward() -> optimizer.step()) runs on this module.  Everything that carries weight or
moves data between rows - every nn.Linear / nn.Conv2d / KPConv contraction, the KPConv neighbour aggregation, attention, the
neighbour max-pool and up-sample gathers - is a `cofii2p_amd.autograd` Function whose forward AND backward are hand-written gfx950
kernels - as are the normalisations over the rows of a map with the activation and residual join behind them (GroupNorm, InstanceNorm,
train-mode BatchNorm: ag.group_norm_act), the L2 normalisations (descriptor rows; Q over the tokens) and the bilinear x2 of the up-samplers.  What is left to
torch's differentiable tensor ops on the same device buffers - the "torch fallback" SURVEY.md row f3 allows - is row-local and
weight-free: LayerNorm, sigmoid, ReLU, concatenations and the 3x3 max-pool of the ResNet stem.  Activations are pixel-major / point-major (rows, C) matrices as in the inference path.

Differences from the inference path, all the reference's own train()-mode semantics:
  * BatchNorm2d of the two ImageUpSample stages uses BATCH statistics and updates running_mean / running_var / num_batches_tracked
    (imagenet.py:381-394 under model.train(), train.py:188);
  * nothing is folded, fused across layers or captured in a hipGraph; ResNet layer3 / layer4 / avg-pool, which feed nothing
    (network.py:87-89) and hold no state, are skipped.
One frame per call (train.py's batch: `torch.squeeze` of a batch of 1).  opt.norm: 'gn' (the shipped configuration), 'bn', 'ln' (`_Norm`).
"""
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import _lib, autograd as ag, ops
from .spec import D_MODEL, DECODERS, ENCODER, GN_GROUPS, LAYER_KINDS, N_HEAD, RESNET_LAYERS

LRELU = 0.1


# ------------------------------------------------------------------------------------------ normalisations over the rows of a map
def group_norm_rows(x, w, b, slope: float = 1.0, res=None, groups: int = GN_GROUPS):
    """modules.py:45-49: nn.GroupNorm over (1, C, N) - statistics over all rows and the channels of a group - with the LeakyReLU and the
    residual join that follow it in the reference's blocks (HIP forward and backward: ag.group_norm_act)."""
    return ag.group_norm_act(x, w, b, groups, slope, res)


def instance_norm_rows(x, slope: float = 1.0, res=None):
    """affine-less nn.InstanceNorm over the positions of one map = per-column normalisation of a (positions, C) matrix (+ ReLU / residual)"""
    return ag.group_norm_act(x, None, None, x.shape[1], slope, res)


def batch_norm_rows(x, P, B, p: str, training: bool, slope: float = 1.0, res=None, eps: float = 1e-5, momentum: float = 0.1):
    """nn.BatchNorm2d on a (1, C, H, W) map = BatchNorm over the rows of (H W, C) (+ ReLU / residual).  train mode: batch statistics, and the
    running buffers move as nn.BatchNorm2d moves them (momentum 0.1, unbiased variance); eval mode: the running statistics, as constants."""
    C = x.shape[1]
    rm, rv = B[p + "running_mean"], B[p + "running_var"]
    if not training:
        fixed = torch.stack([rm, torch.rsqrt(rv + eps)], 1).contiguous()
        return ag.group_norm_act(x, P[p + "weight"], P[p + "bias"], C, slope, res, eps, fixed_stats=fixed)
    y, stats = ag.group_norm_act(x, P[p + "weight"], P[p + "bias"], C, slope, res, eps, return_stats=True)
    with torch.no_grad():
        n = x.shape[0]
        var_b = (1.0 / (stats[:, 1] * stats[:, 1]) - eps).clamp_min(0.0)
        rm.mul_(1.0 - momentum).add_(stats[:, 0], alpha=momentum)
        rv.mul_(1.0 - momentum).add_(var_b * (n / max(n - 1, 1)), alpha=momentum)
        nbt = B.get(p + "num_batches_tracked")
        if nbt is not None:
            nbt.add_(1)
    return y


class _Norm:
    """get_norm() of the point encoder (modules.py:51-60) on (N, C) rows, with the residual join and LeakyReLU that follow it in the
    reference's blocks.  'gn': the GroupNorm wrapper (keys p + "norm.weight"); 'bn': nn.BatchNorm1d over the rows - batch statistics and
    moving buffers while the module trains, running statistics under eval() - on the same HIP kernels as the image branch's BatchNorm2d;
    'ln': nn.LayerNorm over the channels of a row (row-local: torch's differentiable op, as in the point MLP)."""

    def __init__(self, kind: str, P, B, training: bool):
        self.kind, self.P, self.B, self.training = kind, P, B, training

    def __call__(self, p: str, x, slope: float = 1.0, res=None):
        P = self.P
        if self.kind == "gn":
            return group_norm_rows(x, P[p + "norm.weight"], P[p + "norm.bias"], slope, res)
        if self.kind == "bn":
            return batch_norm_rows(x, P, self.B, p, self.training, slope, res)
        y = F.layer_norm(x, (x.shape[1],), P[p + "weight"], P[p + "bias"], 1e-5)
        if res is not None:
            y = y + res
        return F.leaky_relu(y, slope) if slope != 1.0 else y


def pos_sine_table(coords: torch.Tensor) -> torch.Tensor:
    """position_encoding.py:7-50 as a constant (no parameters, inputs need no gradient): (T, n) -> (T, 128)."""
    out = torch.zeros((coords.shape[0], D_MODEL), dtype=torch.float32, device=coords.device)
    return ops.pos_sine(coords.contiguous(), out, accumulate=False)


# ------------------------------------------------------------------------------------------ point encoder (kp_backbone.py:79-128)
def _unary(P, nrm, p, x, relu: bool = True, norm: bool = True, res=None):
    """UnaryBlock (modules.py:63-94): Linear -> get_norm() -> LeakyReLU; `res` joins before the activation (the residual tail, :236-240)"""
    y = ag.linear(x, P[p + "mlp.weight"], P[p + "mlp.bias"])
    if norm:
        return nrm(p + "norm.", y, LRELU if relu else 1.0, res)
    return F.leaky_relu(y, LRELU) if relu else y


def _kpconv(P, B, p, feats, q_pts, s_pts, idx, sigma, tables):
    """kpconv.py:79-122: aggregate, contract with the (15, Cin, Cout) weights, divide by the neighbour count, add the bias."""
    w = P[p + "KPConv.weights"]
    agg, cnt = ag.kpconv_aggregate(feats, q_pts, s_pts, idx, B[p + "KPConv.kernel_points"], sigma, tables)
    w2 = w.permute(2, 0, 1).reshape(w.shape[2], -1)     # (Cout, 15 Cin): column k Cin + c, the aggregate's layout
    return ag.linear(agg, w2, P[p + "KPConv.bias"], rowdiv=cnt)


def _block(P, B, nrm, blk, feats, q_pts, s_pts, idx, tables):
    p = "pc_encoder.%s." % blk.name
    if blk.kind == "conv":   # modules.py:155-159
        y = _kpconv(P, B, p, feats, q_pts, s_pts, idx, blk.sigma, tables)
        return nrm(p + "norm.", y, LRELU)
    x = _unary(P, nrm, p + "unary1.", feats) if blk.cin != blk.mid else feats   # modules.py:222-240
    x = _kpconv(P, B, p, x, q_pts, s_pts, idx, blk.sigma, tables)
    x = nrm(p + "norm_conv.", x, LRELU)
    sc = ag.neighbor_maxpool(feats, idx, tables) if blk.strided else feats
    if blk.has_shortcut_unary:
        sc = _unary(P, nrm, p + "unary_shortcut.", sc, relu=False)
    return _unary(P, nrm, p + "unary2.", x, relu=True, res=sc)   # leaky(norm(unary2) + shortcut)


def kpconv_fpn(P, B, points, neighbors, subsampling, upsampling, feats, tables, kind: str = "gn", training: bool = True, taps=None) -> List[torch.Tensor]:
    """kp_backbone.py:79-128.  taps (optional dict): every encoder block's output by name, for tests"""
    nrm = _Norm(kind, P, B, training)
    x = feats
    stage_out = {}
    for blk in ENCODER:
        st = blk.stage
        if blk.strided:
            q, s, idx = points[st], points[st - 1], subsampling[st - 1]
        else:
            q, s, idx = points[st], points[st], neighbors[st]
        x = _block(P, B, nrm, blk, x, q, s, idx, tables)
        stage_out[st] = x
        if taps is not None:
            taps[blk.name] = x
    s5 = stage_out[4]
    dec = {name: norm for name, _, _, norm in DECODERS}
    l4 = _unary(P, nrm, "pc_encoder.decoder4.", torch.cat([ag.gather_rows(s5, upsampling[3], tables), stage_out[3]], 1), norm=dec["decoder4"])
    l3 = _unary(P, nrm, "pc_encoder.decoder3.", torch.cat([ag.gather_rows(l4, upsampling[2], tables), stage_out[2]], 1), norm=dec["decoder3"])
    l2 = ag.linear(torch.cat([ag.gather_rows(l3, upsampling[1], tables), stage_out[1]], 1), P["pc_encoder.decoder2.mlp.weight"],
                   P["pc_encoder.decoder2.mlp.bias"])
    return [l2, l3, l4, s5]


# ------------------------------------------------------------------------------------------ image encoder (imagenet.py:119-217)
def _nchw(x, H, W):
    return x.t().reshape(1, x.shape[1], H, W)


def _rows(x):
    return x.reshape(x.shape[1], -1).t()


def resnet34_s8(P, img: torch.Tensor):
    """-> (s2, s4, s8) pixel-major maps with their (H, W): the stem, the max-pool and layer1 / layer2 of the ResNet-34."""
    p = "img_encoder.backbone."
    col, H, W = ops.im2col_stem(img.contiguous())                  # (Ho Wo, 160): 7 x 7 x 3 = 147 columns, zero padded
    w = P[p + "conv1.weight"]
    w2 = F.pad(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), (0, col.shape[1] - 147))
    x = instance_norm_rows(ag.linear(col, w2), 0.0)
    outs = [(x, H, W)]
    x4 = F.max_pool2d(_nchw(x, H, W), 3, 2, 1)
    H, W = x4.shape[2:]
    x = _rows(x4)
    for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS[:2], start=1):
        for b in range(blocks):
            q = "%slayer%d.%d." % (p, li, b)
            st = stride if b == 0 else 1
            y, Ho, Wo = ag.conv2d(x, H, W, P[q + "conv1.weight"], st)
            y = instance_norm_rows(y, 0.0)
            y, _, _ = ag.conv2d(y, Ho, Wo, P[q + "conv2.weight"], 1)
            if (q + "downsample.0.weight") in P:
                d, _, _ = ag.conv2d(x, H, W, P[q + "downsample.0.weight"], st, pad=0)
                x = instance_norm_rows(y, 0.0, res=instance_norm_rows(d))
            else:
                x = instance_norm_rows(y, 0.0, res=x.contiguous())
            H, W = Ho, Wo
        outs.append((x, H, W))
    return outs


def _residual_conv(P, B, p, x, H, W, training):
    """imagenet.py:377-411."""
    # conv_skip and conv1 read the same input (imagenet.py:399-402): one unfolded operand, one contraction with the stacked filters
    ws, w1 = P[p + "conv_skip.0.weight"], P[p + "conv1.weight"]
    Cout = ws.shape[0]
    both = ag.linear(ag.im2col(x.contiguous(), H, W, 3, 1, 1), torch.cat([ws, w1], 0).permute(0, 2, 3, 1).reshape(2 * Cout, -1))
    identity = batch_norm_rows(both[:, :Cout], P, B, p + "conv_skip.1.", training)
    out = batch_norm_rows(both[:, Cout:], P, B, p + "bn1.", training, 0.0)
    return batch_norm_rows(ag.conv2d(out, H, W, P[p + "conv2.weight"])[0], P, B, p + "bn2.", training, 0.0, res=identity)


def image_upsample(P, B, name, low, h, w, skip, training):
    """imagenet.py:431-444: bilinear x2 (align_corners=False), concat with the skip map, two ResidualConv."""
    x = ag.upsample2x_cat(low, skip, h, w)
    x = _residual_conv(P, B, name + ".conv.0.", x, 2 * h, 2 * w, training)
    return _residual_conv(P, B, name + ".conv.1.", x, 2 * h, 2 * w, training)


# ------------------------------------------------------------------------------------------ transformer (transformer.py:43-104)
def loftr_layer(P, p, x, src, nhead: int = N_HEAD):
    C = x.shape[1]
    if src is x:   # self layer: the three projections read the same tokens - one contraction with the stacked weights [Wq; Wk; Wv]
        qkv = ag.linear(x, torch.cat([P[p + "q_proj.weight"], P[p + "k_proj.weight"], P[p + "v_proj.weight"]], 0))
        q, k, v = qkv.split(C, 1)   # (split, not three slices: its backward is one concatenation)
    else:
        q = ag.linear(x, P[p + "q_proj.weight"])
        kv = ag.linear(src, torch.cat([P[p + "k_proj.weight"], P[p + "v_proj.weight"]], 0))
        k, v = kv.split(C, 1)
    q = ag.normalize_cols(q)    # transformer.py:53: F.normalize's default dim=1 on (1, L, H, D) = over the L tokens
    msg = ag.attention(q, k, v, nhead)
    msg = F.layer_norm(ag.linear(msg, P[p + "merge.weight"]), (C,), P[p + "norm1.weight"], P[p + "norm1.bias"])
    h = ag.linear(F.relu(ag.linear(torch.cat([x, msg], 1), P[p + "mlp.0.weight"])), P[p + "mlp.2.weight"])
    return x + F.layer_norm(h, (C,), P[p + "norm2.weight"], P[p + "norm2.bias"])


def transformer(P, tok_img, tok_pc):
    for l, kind in enumerate(LAYER_KINDS):
        p = "transformer.layers.%d." % l
        if kind == "self":
            tok_img = loftr_layer(P, p, tok_img, tok_img)
            tok_pc = loftr_layer(P, p, tok_pc, tok_pc)
        else:   # the point stream attends to the ALREADY UPDATED image stream (transformer.py:99-100)
            tok_img = loftr_layer(P, p, tok_img, tok_pc)
            tok_pc = loftr_layer(P, p, tok_pc, tok_img)
    return tok_img, tok_pc


def score_head(P, head, tokens):
    """network.py:42-43 on token-major data."""
    w0, w3, w6 = (P["%s.%d.weight" % (head, i)] for i in (0, 3, 6))
    y = instance_norm_rows(ag.linear(tokens, w0.reshape(w0.shape[0], -1)), 0.0)
    y = instance_norm_rows(ag.linear(y, w3.reshape(w3.shape[0], -1)), 0.0)
    return torch.sigmoid(ag.linear(y, w6.reshape(w6.shape[0], -1)))   # (T, 1)


def pc_feature_mlp(P, x):
    """network.py:29."""
    p = "pc_feature_layer."
    x = F.relu(F.layer_norm(ag.linear(x, P[p + "0.weight"]), (1024,), P[p + "1.weight"], P[p + "1.bias"]))
    x = F.relu(F.layer_norm(ag.linear(x, P[p + "3.weight"]), (512,), P[p + "4.weight"], P[p + "4.bias"]))
    return ag.linear(x, P[p + "6.weight"])


OVERLAP_BRANCHES = True   # image branch on a side stream while a hipGraph records (forward_train)
_SIDE = {}


def _side_stream(dev) -> torch.cuda.Stream:
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


# ------------------------------------------------------------------------------------------ network.py:74-164, train / val branch
def forward_train(model, pc_data_dict: Dict, img: torch.Tensor, fine_center_kpt_coors: torch.Tensor, fine_pc_inline_index: torch.Tensor):
    """-> the reference's 8-tuple (fine_center_xy = coarse_pc_points = None) with a graph behind every tensor."""
    if not img.is_cuda:
        raise _lib.CofiError("CoFiI2P.forward needs CUDA (HIP) tensors: there is no CPU path")
    if img.dim() != 4 or img.shape[0] != 1:
        raise ValueError("training runs one frame per forward (train.py squeezes a batch of 1)")
    _lib.load()
    P = dict(model.named_parameters())
    B = dict(model.named_buffers())
    training = model.training
    as32 = model._as_idx32
    points = [p.contiguous() for p in pc_data_dict["points"]]
    neighbors = [as32(t) for t in pc_data_dict["neighbors"]]
    subsampling = [as32(t) for t in pc_data_dict["subsampling"]]
    upsampling = [as32(t) for t in pc_data_dict["upsampling"]]
    feats = pc_data_dict["feats"].contiguous()
    tables = ag.TableCache()

    # While a hipGraph records (train_step.GraphedTrainStep) the image branch - ResNet, both up-samplers, the patches - is issued on a side
    # stream: forked here, joined where its results meet the point branch.  The recording then holds two independent chains (autograd
    # runs every backward node on the stream of its forward), and the replay overlaps their many small kernels.  Eager execution is bound
    # by the issuing thread and stays on one stream.
    dev = img.device
    main = torch.cuda.current_stream(dev)
    fork = torch.cuda.is_current_stream_capturing() and OVERLAP_BRANCHES
    side = _side_stream(dev) if fork else main
    if fork:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        (s2, H2, W2), (s4, H4, W4), (s8, H8, W8) = resnet34_s8(P, img)
        s8n = ag.normalize_rows(s8)                                              # network.py:90
        tok_img = s8n + pos_sine_table(model._pixel_grid(H8, W8, 1, dev))         # network.py:104-110
        tokens_ready = torch.cuda.Event()
        tokens_ready.record(side)
        up4 = image_upsample(P, B, "img_upsample_1", s8n, H8, W8, s4, training)
        up2 = ag.normalize_rows(image_upsample(P, B, "img_upsample_2", up4, H4, W4, s2, training))   # (H2 W2, 64)
        # network.py:137-141: 4 x 4 patches around the labelled pixels
        ctr = fine_center_kpt_coors.to(device=dev)
        lt = torch.floor(ctr.to(torch.float32) - 2.0).to(torch.int64)             # network.py:213: left/top = floor(centre - size / 2)
        ar = torch.arange(4, device=dev)
        rows, cols = lt[1][:, None] + ar[None], lt[0][:, None] + ar[None]         # (K, 4)
        # (a host read: not possible while a hipGraph records - GraphedTrainStep checks the frame before it launches the recording)
        if not torch.cuda.is_current_stream_capturing() and (bool(((rows < 0) | (rows >= H2)).any()) or bool(((cols < 0) | (cols >= W2)).any())):
            raise AssertionError("patch leaves the feature map (network.py:222)")
        pix = (rows[:, :, None] * W2 + cols[:, None, :]).reshape(-1)             # (K 16,)
        patches = ag.gather_rows(up2, pix.to(torch.int32), tables).reshape(ctr.shape[1], 4, 4, -1).permute(0, 3, 1, 2)

    pc_set = kpconv_fpn(P, B, points, neighbors, subsampling, upsampling, feats, tables, model.pc_norm_kind, training)
    fine_pc = ag.normalize_rows(pc_set[0])                                   # network.py:83
    pc_mid = ag.normalize_rows(pc_feature_mlp(P, pc_set[-1]))                # network.py:84
    tok_pc = pc_mid + pos_sine_table(points[-1])                              # network.py:107,111
    fine_feat = ag.gather_rows(fine_pc, as32(fine_pc_inline_index.reshape(-1)), tables)   # network.py:137: descriptors of the labelled points
    if fork:
        main.wait_event(tokens_ready)
        tok_img.record_stream(main)
    tok_img, tok_pc = transformer(P, tok_img, tok_pc)
    pc_score = score_head(P, "pc_score_layer", tok_pc)
    img_score = score_head(P, "img_score_layer", tok_img)
    pc_desc = ag.normalize_rows(tok_pc).t()                                  # (C, N4)  network.py:125
    img_desc = ag.normalize_rows(tok_img).t().reshape(1, D_MODEL, H8, W8)    # network.py:126
    if fork:
        main.wait_stream(side)
        patches.record_stream(main)
    N4 = points[-1].shape[0]
    return (img_desc, pc_desc, img_score.reshape(1, 1, H8, W8), pc_score.reshape(1, 1, N4), patches, fine_feat, None, None)
