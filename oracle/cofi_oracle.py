"""CPU ORACLE — test infrastructure, NOT the product.

A functional fp32 restatement (torch-CPU / numpy) of the CoFiI2P forward hot path
(SURVEY.md §8a rows a-1 … a-18).  Every function cites the reference file:line it follows
(paths relative to the reference root).  It is pinned against the reference itself:
`tests/tools/make_golden.py` imports /root/reference in the development container, runs both on
the same seeded inputs + name-keyed weights and commits the reference's outputs under
`tests/golden/` (`tests/test_oracle_golden.py` re-checks the oracle against them everywhere).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product (`cofii2p_amd/`) never does: it has no CPU fallback.

All tensors are torch.float32 on CPU; indices int64; ``sd`` is a ``{name: tensor}`` state_dict
with the reference's key names.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# a-1  brute-force KNN pyramid (model/kpconv/preprocess_data.py:109-143, 145-203)
# --------------------------------------------------------------------------------------


def expansion_sqdist(query: Tensor, support: Tensor) -> Tensor:
    """``-2 q·sᵀ + |q|² + |s|²`` clamped at 1e-12 (preprocess_data.py:120-128,
    network.py:239-246).  (Q,3),(S,3) -> (Q,S)."""
    d = -2.0 * (query @ support.t())
    d = d + (query * query).sum(-1)[:, None]
    d = d + (support * support).sum(-1)[None, :]
    return d.clamp_min(1e-12)


def knn_torch(support: Tensor, query: Tensor, k: int) -> Tensor:
    """preprocess_data.py:131-143 ``knn(nodes, points, k)``: for every query row the k support
    rows of smallest expansion distance, ascending.  Tie order is whatever torch.topk gives;
    the tie-defined variant used for bit-exact tests is `knn_oracle.c`."""
    out = torch.empty((query.shape[0], k), dtype=torch.int64)
    step = 2048
    for i in range(0, query.shape[0], step):
        d = expansion_sqdist(query[i : i + step], support)
        out[i : i + step] = d.topk(k, dim=-1, largest=False)[1]
    return out


def subsample_half(points_3xn: np.ndarray, rng: np.random.RandomState) -> np.ndarray:
    """preprocess_data.py:55-59: N//2 random columns WITH replacement."""
    n = points_3xn.shape[1]
    pick = rng.choice(np.arange(n), size=n // 2)
    return points_3xn[:, pick]


def build_pyramid(points_3xn: np.ndarray, num_stages: int, rng: np.random.RandomState, k: int = 128, knn=knn_torch):
    """preprocess_data.py:145-203 (`precompute_point_cloud_cuda`, the importable twin of the
    open3d-based `precompute_point_cloud_stack_mode` :36-107): points / neighbors / subsampling /
    upsampling lists."""
    pts_np = points_3xn
    points: List[Tensor] = []
    lengths: List[int] = []
    for i in range(num_stages):
        if i > 0:
            pts_np = subsample_half(pts_np, rng)
        points.append(torch.from_numpy(np.ascontiguousarray(pts_np.T)).float())
        lengths.append(points[-1].shape[0])
    neighbors, subsampling, upsampling = [], [], []
    for i in range(num_stages):
        cur = points[i]
        neighbors.append(knn(cur, cur, k))
        if i < num_stages - 1:
            sub = points[i + 1]
            subsampling.append(knn(cur, sub, k))  # (N_{i+1}, k) into stage i
            upsampling.append(knn(sub, cur, k))  # (N_i, k) into stage i+1
    return {"points": points, "lengths": lengths, "neighbors": neighbors, "subsampling": subsampling, "upsampling": upsampling}


# --------------------------------------------------------------------------------------
# a-2 … a-6  KPConv-FPN
# --------------------------------------------------------------------------------------


def _rows_with_pad(x: Tensor, pad_value: float = 0.0) -> Tensor:
    pad = torch.full((1, x.shape[1]), pad_value, dtype=x.dtype)
    return torch.cat([x, pad], 0)


def kpconv(s_feats: Tensor, q_points: Tensor, s_points: Tensor, idx: Tensor, kernel_points: Tensor, weights: Tensor,
           bias: Optional[Tensor], sigma: float, chunk: int = 1024) -> Tensor:
    """kpconv.py:79-122.  Shadow index ``idx == N`` selects a point at +1e6 / a zero feature row.
    Processed in row chunks to bound memory; every row is independent."""
    sp = _rows_with_pad(s_points, 1e6)
    sf = _rows_with_pad(s_feats, 0.0)
    K, cin, cout = weights.shape
    w2 = weights.reshape(K * cin, cout)
    out = torch.empty((q_points.shape[0], cout), dtype=torch.float32)
    for a in range(0, q_points.shape[0], chunk):
        ii = idx[a : a + chunk]
        rel = sp[ii] - q_points[a : a + chunk, None, :]  # (m,H,3)
        diff = rel[:, :, None, :] - kernel_points[None, None]  # (m,H,K,3)
        infl = (1.0 - diff.pow(2).sum(-1).sqrt() / sigma).clamp_min(0.0)  # (m,H,K)
        nf = sf[ii]  # (m,H,cin)
        agg = infl.transpose(1, 2) @ nf  # (m,K,cin)
        o = agg.reshape(agg.shape[0], K * cin) @ w2  # == sum_k agg[:,k] @ W[k]
        cnt = (nf.sum(-1) > 0.0).sum(-1).clamp_min(1)  # kpconv.py:113-115
        out[a : a + chunk] = o / cnt[:, None].to(torch.float32)
    if bias is not None:
        out = out + bias
    return out


def group_norm_rows(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 32, eps: float = 1e-5) -> Tensor:
    """modules.py:32-49: nn.GroupNorm on (1,C,N) — statistics over ALL rows x C/groups channels."""
    n, c = x.shape
    g = x.reshape(n, groups, c // groups)
    var, mean = torch.var_mean(g, dim=(0, 2), unbiased=False, keepdim=True)
    y = ((g - mean) * torch.rsqrt(var + eps)).reshape(n, c)
    return y * gamma + beta


def leaky(x: Tensor) -> Tensor:
    return F.leaky_relu(x, 0.1)


def pc_norm(sd, p: str, x: Tensor) -> Tensor:
    """get_norm() of modules.py:51-60 applied to (N, C) rows, picked by the keys the state_dict holds under prefix p ("...norm." /
    "...norm_conv."): GroupNorm wrapper (p + "norm.weight"), BatchNorm1d (running statistics; batch statistics + buffer update when the module trains), LayerNorm."""
    if (p + "norm.weight") in sd:
        return group_norm_rows(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
    if (p + "running_mean") in sd:
        return _bn_eval(sd, p, x)   # nn.BatchNorm1d over the rows: running statistics, or batch statistics under forward(train_bn=True)
    return F.layer_norm(x, (x.shape[1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def unary_block(sd, p: str, x: Tensor, norm: bool = True, act: bool = True) -> Tensor:
    """modules.py:63-112 UnaryBlock / LastUnaryBlock."""
    y = x @ sd[p + "mlp.weight"].t() + sd[p + "mlp.bias"]
    if norm:
        y = pc_norm(sd, p + "norm.", y)
        if act:
            y = leaky(y)
    return y


def neighbor_maxpool(x: Tensor, idx: Tensor) -> Tensor:
    """functional.py:53-66."""
    return _rows_with_pad(x)[idx].max(1)[0]


def nearest_upsample(x: Tensor, idx: Tensor) -> Tensor:
    """functional.py:5-21 — column 0 only."""
    return _rows_with_pad(x)[idx[:, 0]]


def conv_block(sd, p: str, feats, q_pts, s_pts, idx, sigma) -> Tensor:
    """modules.py:115-159."""
    y = kpconv(feats, q_pts, s_pts, idx, sd[p + "KPConv.kernel_points"], sd[p + "KPConv.weights"], sd[p + "KPConv.bias"], sigma)
    return leaky(pc_norm(sd, p + "norm.", y))


def residual_block(sd, p: str, feats, q_pts, s_pts, idx, sigma, strided: bool) -> Tensor:
    """modules.py:162-240 (bottleneck: unary1 -> KPConv -> GN -> LReLU -> unary2 (+ shortcut) -> LReLU)."""
    x = unary_block(sd, p + "unary1.", feats) if (p + "unary1.mlp.weight") in sd else feats
    x = kpconv(x, q_pts, s_pts, idx, sd[p + "KPConv.kernel_points"], sd[p + "KPConv.weights"], sd[p + "KPConv.bias"], sigma)
    x = leaky(pc_norm(sd, p + "norm_conv.", x))
    x = unary_block(sd, p + "unary2.", x, act=False)
    sc = neighbor_maxpool(feats, idx) if strided else feats
    if (p + "unary_shortcut.mlp.weight") in sd:
        sc = unary_block(sd, p + "unary_shortcut.", sc, act=False)
    return leaky(x + sc)


# (name, kind, query stage, strided, sigma multiplier) — kp_backbone.py:11-73
_ENC = (
    ("encoder1_1", "conv", 0, False, 1), ("encoder1_2", "res", 0, False, 1),
    ("encoder2_1", "res", 1, True, 1), ("encoder2_2", "res", 1, False, 2), ("encoder2_3", "res", 1, False, 2),
    ("encoder3_1", "res", 2, True, 2), ("encoder3_2", "res", 2, False, 4), ("encoder3_3", "res", 2, False, 4),
    ("encoder4_1", "res", 3, True, 4), ("encoder4_2", "res", 3, False, 8), ("encoder4_3", "res", 3, False, 8),
    ("encoder5_1", "res", 4, True, 8), ("encoder5_2", "res", 4, False, 16), ("encoder5_3", "res", 4, False, 16),
)


def kpconv_fpn(sd, data: Dict, init_sigma: float = 0.2, taps: Optional[dict] = None) -> List[Tensor]:
    """kp_backbone.py:79-128.  Returns [s2-latent (N1,64), s3 (N2,512), s4 (N3,1024), s5 (N4,2048)]."""
    pts, nb, sub, up = data["points"], data["neighbors"], data["subsampling"], data["upsampling"]
    x = data["feats"]
    stage_out = {}
    for name, kind, st, strided, mult in _ENC:
        p = "pc_encoder.%s." % name
        if strided:
            q, s, idx = pts[st], pts[st - 1], sub[st - 1]
        else:
            q, s, idx = pts[st], pts[st], nb[st]
        if kind == "conv":
            x = conv_block(sd, p, x, q, s, idx, init_sigma * mult)
        else:
            x = residual_block(sd, p, x, q, s, idx, init_sigma * mult, strided)
        stage_out[st] = x
        if taps is not None:
            taps[name] = x
    s5 = stage_out[4]
    l4 = unary_block(sd, "pc_encoder.decoder4.", torch.cat([nearest_upsample(s5, up[3]), stage_out[3]], 1))
    l3 = unary_block(sd, "pc_encoder.decoder3.", torch.cat([nearest_upsample(l4, up[2]), stage_out[2]], 1))
    l2 = unary_block(sd, "pc_encoder.decoder2.", torch.cat([nearest_upsample(l3, up[1]), stage_out[1]], 1), norm=False)
    return [l2, l3, l4, s5]


# --------------------------------------------------------------------------------------
# a-7, a-8  image branch
# --------------------------------------------------------------------------------------


def _inorm(x: Tensor) -> Tensor:
    return F.instance_norm(x, eps=1e-5)


def resnet34_in(sd, img: Tensor) -> List[Tensor]:
    """imagenet.py:196-217 with norm_layer = InstanceNorm2d (no affine) (:123)."""
    p = "img_encoder.backbone."
    x = F.relu(_inorm(F.conv2d(img, sd[p + "conv1.weight"], stride=2, padding=3)))
    outs = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for b in range(blocks):
            q = "%slayer%d.%d." % (p, li, b)
            st = stride if b == 0 else 1
            y = F.relu(_inorm(F.conv2d(x, sd[q + "conv1.weight"], stride=st, padding=1)))
            y = _inorm(F.conv2d(y, sd[q + "conv2.weight"], padding=1))
            if (q + "downsample.0.weight") in sd:
                x = _inorm(F.conv2d(x, sd[q + "downsample.0.weight"], stride=st))
            x = F.relu(y + x)
        outs.append(x)
    outs.append(F.adaptive_avg_pool2d(x, 1))
    return outs


BN_TRAIN = False   # set by forward(..., train_bn=True): the module under model.train() (train.py:188)


def _bn_eval(sd, p: str, x: Tensor) -> Tensor:
    """nn.BatchNorm2d: running statistics (eval), or - forward(train_bn=True) - batch statistics with the running buffers of `sd` updated in
    place (momentum 0.1, unbiased variance), as imagenet.py:381-394 behaves under model.train()."""
    if BN_TRAIN:
        nbt = sd.get(p + "num_batches_tracked")
        if nbt is not None:
            nbt.add_(1)
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def residual_conv(sd, p: str, x: Tensor) -> Tensor:
    """imagenet.py:377-411 (BatchNorm in eval mode = running statistics)."""
    skip = _bn_eval(sd, p + "conv_skip.1.", F.conv2d(x, sd[p + "conv_skip.0.weight"], padding=1))
    y = F.relu(_bn_eval(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], padding=1)))
    y = _bn_eval(sd, p + "bn2.", F.conv2d(y, sd[p + "conv2.weight"], padding=1))
    return F.relu(y + skip)


def image_upsample(sd, p: str, low: Tensor, skip: Tensor) -> Tensor:
    """imagenet.py:431-444."""
    x = torch.cat([F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=False), skip], 1)
    return residual_conv(sd, p + "conv.1.", residual_conv(sd, p + "conv.0.", x))


# --------------------------------------------------------------------------------------
# a-10 … a-13  position embedding + transformer
# --------------------------------------------------------------------------------------


def pos_sine(xyz: Tensor, d_model: int = 128, temperature: float = 10000.0) -> Tensor:
    """position_encoding.py:7-50.  xyz (..., n_dim) -> (..., d_model)."""
    n_dim = xyz.shape[-1]
    f = d_model // n_dim // 2 * 2
    i = torch.arange(f, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / f)
    ang = (xyz * (2 * math.pi)).unsqueeze(-1) / dim_t
    emb = torch.stack([ang[..., 0::2].sin(), ang[..., 1::2].cos()], -1).reshape(*xyz.shape[:-1], -1)
    return F.pad(emb, (0, d_model - f * n_dim))


def full_attention(q: Tensor, k: Tensor, v: Tensor, chunk: int = 2048) -> Tensor:
    """linear_attention.py:56-79.  (L,H,D),(S,H,D),(S,H,D) -> (L,H,D).  The softmax runs over the whole key axis per query row,
    so evaluating the query rows in chunks changes no value: the stress frame's (4, 22400, 22400) score tensor (8 GB in the
    reference) then never has to exist on the host."""
    out = []
    for l0 in range(0, q.shape[0], chunk):
        scores = torch.einsum("lhd,shd->hls", q[l0:l0 + chunk], k) / math.sqrt(q.shape[-1])
        out.append(torch.einsum("hls,shd->lhd", scores.softmax(-1), v))
    return out[0] if len(out) == 1 else torch.cat(out, 0)


def loftr_layer(sd, p: str, x: Tensor, src: Tensor, nhead: int = 4) -> Tensor:
    """transformer.py:43-64.  x (L,C), src (S,C).  NOTE F.normalize's default dim=1 on the
    (1,L,H,D) view normalises every channel over the L TOKENS (transformer.py:53)."""
    L, C = x.shape
    q = x @ sd[p + "q_proj.weight"].t()
    q = q / q.norm(dim=0, keepdim=True).clamp_min(1e-12)
    k = src @ sd[p + "k_proj.weight"].t()
    v = src @ sd[p + "v_proj.weight"].t()
    d = C // nhead
    msg = full_attention(q.view(L, nhead, d), k.view(-1, nhead, d), v.view(-1, nhead, d)).reshape(L, C)
    msg = F.layer_norm(msg @ sd[p + "merge.weight"].t(), (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = F.relu(torch.cat([x, msg], 1) @ sd[p + "mlp.0.weight"].t()) @ sd[p + "mlp.2.weight"].t()
    return x + F.layer_norm(h, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])


def transformer(sd, f_img: Tensor, f_pc: Tensor, kinds: Sequence[str] = ("self", "cross") * 4, taps: Optional[dict] = None):
    """transformer.py:85-104: shared weights for both streams; in a cross layer the point stream
    attends to the ALREADY UPDATED image stream (:99-100)."""
    for l, kind in enumerate(kinds):
        p = "transformer.layers.%d." % l
        if kind == "self":
            f_img = loftr_layer(sd, p, f_img, f_img)
            f_pc = loftr_layer(sd, p, f_pc, f_pc)
        else:
            f_img = loftr_layer(sd, p, f_img, f_pc)
            f_pc = loftr_layer(sd, p, f_pc, f_img)
        if taps is not None:
            taps["layer%d" % l] = (f_img, f_pc)
    return f_img, f_pc


# --------------------------------------------------------------------------------------
# a-9, a-14 … a-18  heads and matching
# --------------------------------------------------------------------------------------


def pc_feature_mlp(sd, x: Tensor) -> Tensor:
    """network.py:29 Linear-LN-ReLU-Linear-LN-ReLU-Linear (no bias)."""
    p = "pc_feature_layer."
    x = F.relu(F.layer_norm(x @ sd[p + "0.weight"].t(), (1024,), sd[p + "1.weight"], sd[p + "1.bias"]))
    x = F.relu(F.layer_norm(x @ sd[p + "3.weight"].t(), (512,), sd[p + "4.weight"], sd[p + "4.bias"]))
    return x @ sd[p + "6.weight"].t()


def score_head(sd, p: str, tokens: Tensor) -> Tensor:
    """network.py:42-43: 1x1 conv -> InstanceNorm -> ReLU (x2) -> 1x1 conv -> sigmoid, written on
    token-major (T,C) data: an InstanceNorm over the positions of one channel is a per-column
    normalisation.  Returns (T,)."""
    def inorm_cols(y):
        var, mean = torch.var_mean(y, dim=0, unbiased=False, keepdim=True)
        return (y - mean) * torch.rsqrt(var + 1e-5)

    w0 = sd[p + "0.weight"].reshape(128, 128)
    w3 = sd[p + "3.weight"].reshape(64, 128)
    w6 = sd[p + "6.weight"].reshape(1, 64)
    y = F.relu(inorm_cols(tokens @ w0.t()))
    y = F.relu(inorm_cols(y @ w3.t()))
    return torch.sigmoid(y @ w6.t())[:, 0]


def score_thresholds(n: int = 64) -> List[float]:
    """network.py:147-151: python-float ``thrs = 0.9; thrs -= 0.02`` sequence."""
    out, t = [], 0.9
    for _ in range(n):
        out.append(t)
        t -= 0.02
    return out


def fine_process(score: Tensor, pc_desc_cn: Tensor, img_desc_chw: Tensor, thr: float) -> Tuple[Tensor, Tensor]:
    """network.py:167-187.  score (N,), pc_desc (C,N), img_desc (C,H,W) -> coarse_xy (2,n) float
    [x = column, y = row], selected point indices (n,) ascending."""
    C, H, W = img_desc_chw.shape
    sel = torch.where(score >= thr)[0]
    img_flat = img_desc_chw.reshape(C, H * W)
    dist = 1.0 - (img_flat.unsqueeze(-1) * pc_desc_cn[:, sel].unsqueeze(-2)).sum(0)  # (HW, n)
    pix = dist.argmin(0)
    xy = torch.stack([(pix % W).float(), (pix // W).float()], 0)
    keep = (xy[0] >= 2) & (xy[0] <= 62) & (xy[1] <= 18) & (xy[1] >= 2)  # hard-coded KITTI borders (:184)
    return xy[:, keep], sel[keep]


def point2node(nodes: Tensor, points: Tensor) -> Tensor:
    """network.py:250-264: index of the nearest node for each point (expansion distance)."""
    return expansion_sqdist(points, nodes).topk(1, dim=-1, largest=False)[1].squeeze(-1)


def extract_patch(fmap_chw: Tensor, centers_xy: Tensor, size: int = 4) -> Tensor:
    """network.py:206-226: (C,H,W), (2,n) -> (n,C,size,size); window [c-size/2, c+size/2)."""
    lt = torch.floor(centers_xy - size / 2).long()
    ar = torch.arange(size)
    rows = lt[1][:, None] + ar[None]  # (n,size)
    cols = lt[0][:, None] + ar[None]
    if (rows < 0).any() or (cols < 0).any() or (rows >= fmap_chw.shape[1]).any() or (cols >= fmap_chw.shape[2]).any():
        raise AssertionError("patch leaves the feature map (network.py:222)")
    return fmap_chw[:, rows[:, :, None], cols[:, None, :]].permute(1, 0, 2, 3).contiguous()


def fine_match(patches_nc16: Tensor, pc_feats_nc: Tensor, center_xy: Tensor) -> Tuple[Tensor, Tensor]:
    """evaluation/eval_all.py:99-105 (also train.py:272-278).  Keeps the reference's x/y swap:
    x += idx // 4, y += idx % 4.  Returns fine_xy (2,n), argmax index (n,)."""
    sim = torch.cosine_similarity(patches_nc16.unsqueeze(-1), pc_feats_nc.unsqueeze(-1).unsqueeze(-2), dim=1).squeeze(-1)
    best = sim.argmax(1)
    xy = center_xy - 2
    xy = torch.stack([xy[0] + best // 4, xy[1] + best % 4], 0)
    return xy, best


def get_P_diff(P_pred: np.ndarray, P_gt: np.ndarray) -> Tuple[float, float]:
    """evaluation/eval_all.py:16-22: RTE = |t| of inv(P_pred)·P_gt, RRE = sum |euler xzy| (deg)."""
    from scipy.spatial.transform import Rotation

    d = np.linalg.inv(P_pred) @ P_gt
    ang = Rotation.from_matrix(d[:3, :3]).as_euler("xzy", degrees=True)
    return float(np.linalg.norm(d[:3, 3])), float(np.sum(np.abs(ang)))


# --------------------------------------------------------------------------------------
# a-14  CoFiI2P.forward (network.py:74-164)
# --------------------------------------------------------------------------------------


def forward(sd, data: Dict, img: Tensor, fine_center_kpt_coors: Optional[Tensor], fine_pc_inline_index: Optional[Tensor],
            mode: str, taps: Optional[dict] = None, train_bn: bool = False):
    """Returns the reference's 8-tuple.  ``img`` is (1,3,H,W).  train_bn: the up-sampler's BatchNorm on batch statistics (the module in
    train() mode); every function here is written with differentiable torch ops, so torch.autograd through this forward is the
    oracle of the backward (row f3)."""
    global BN_TRAIN
    saved, BN_TRAIN = BN_TRAIN, bool(train_bn)
    try:
        return _forward(sd, data, img, fine_center_kpt_coors, fine_pc_inline_index, mode, taps)
    finally:
        BN_TRAIN = saved


def _forward(sd, data: Dict, img: Tensor, fine_center_kpt_coors: Optional[Tensor], fine_pc_inline_index: Optional[Tensor],
             mode: str, taps: Optional[dict] = None):
    pc_set = kpconv_fpn(sd, data, taps=taps)
    img_set = resnet34_in(sd, img)
    fine_pc = F.normalize(pc_set[0], dim=1)  # (N1,64)
    pc_mid = F.normalize(pc_feature_mlp(sd, pc_set[-1]), dim=1)  # (N4,128)
    s2, s4 = img_set[0], img_set[1]
    s8 = F.normalize(img_set[2], dim=1)
    _, C, H8, W8 = s8.shape
    gy, gx = torch.meshgrid(torch.arange(H8), torch.arange(W8), indexing="ij")
    grid = torch.stack([gy, gx], -1).reshape(H8 * W8, 2)  # (row, col) — network.py:104-105
    tok_img = s8[0].reshape(C, H8 * W8).t() + pos_sine(grid)
    tok_pc = pc_mid + pos_sine(data["points"][-1])
    if taps is not None:
        taps["tok_img"], taps["tok_pc"] = tok_img, tok_pc
    tok_img, tok_pc = transformer(sd, tok_img, tok_pc, taps=taps)
    pc_score = score_head(sd, "pc_score_layer.", tok_pc)
    img_score = score_head(sd, "img_score_layer.", tok_img)
    pc_desc = F.normalize(tok_pc.t(), dim=0)  # (128,N4)
    img_mid = tok_img.t().reshape(1, C, H8, W8)
    img_desc = F.normalize(img_mid, dim=1)
    up4 = image_upsample(sd, "img_upsample_1.", s8, s4)
    up2 = F.normalize(image_upsample(sd, "img_upsample_2.", up4, s2), dim=1)  # (1,64,H2,W2)
    if taps is not None:
        taps["up2"] = up2
        taps["fine_pc"] = fine_pc

    if mode in ("train", "val"):
        fine_pc_feat = fine_pc[fine_pc_inline_index]
        patches = extract_patch(up2[0], fine_center_kpt_coors)
        center_xy, coarse_pts = None, None
    elif mode == "test":
        sel = None
        for thr in score_thresholds():
            xy, sel = fine_process(pc_score, pc_desc, img_desc[0], float(np.float32(thr)))
            if sel.numel() >= 4:
                break
        else:
            raise RuntimeError("fewer than 4 coarse matches at every threshold")
        coarse_pts = data["points"][-1][sel]
        node = point2node(data["points"][1], coarse_pts)
        center_xy = xy * 4
        patches = extract_patch(up2[0], center_xy).reshape(-1, up2.shape[1], 16)
        fine_pc_feat = fine_pc[node]
    else:
        raise ValueError(mode)
    return (img_desc, pc_desc, img_score.reshape(1, 1, H8, W8), pc_score.reshape(1, 1, -1), patches, fine_pc_feat,
            center_xy, coarse_pts)
