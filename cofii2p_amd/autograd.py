"""torch.autograd Functions over the HIP kernels: the differentiable operators of the training path.

Forward AND backward of every operator here run hand-written gfx950 kernels through the C ABI; torch carries the graph edges and the
saved tensors.  The dense contractions of the backward are the forward's GEMM on transposed operands (`ops.transpose` + `ops.gemm`);
the adjoints of the neighbour gathers walk the TRANSPOSED index table (`TransposedTable`, CSR) so that every gradient row is summed
in a fixed order - no float atomics, bit-reproducible (csrc/backward.hip).

    linear(x, w, bias, rowdiv)                nn.Linear / 1x1 convolution / KPConv part 2  (y = x w^T / rowdiv + bias)
    kpconv_aggregate(feats, q, s, idx, ...)   model/kpconv/kpconv.py:91-105 -> (agg (M, 15 C), neighbour count)
    neighbor_maxpool(x, idx), gather_rows     model/kpconv/functional.py:53-66, 5-21
    im2col(x, H, W, ks, stride, pad)          unfolded operand of a convolution of the image branch (conv = im2col + linear)
    attention(q, k, v, nhead)                 model/transformer/linear_attention.py:56-79
    normalize_rows(x)                         F.normalize(x, dim=1): network.py:83-84, 90, 125-126
    normalize_cols(x)                         F.normalize(x, dim=0): transformer.py:53 normalises Q over the tokens
    upsample2x_cat(low, skip, h, w)           imagenet.py:433-434: bilinear x2 + concatenation with the skip map
    group_norm_act(x, gamma, beta, groups, slope, res)   GroupNorm / InstanceNorm / train-mode BatchNorm over the rows + activation + residual
"""
import math
from typing import Optional

import torch

from . import _lib, ops
from .ops import _ld, _mat, _p, _stream


# ------------------------------------------------------------------------------------------ transposed index tables
class TransposedTable:
    """CSR transpose of an index table idx (M, H) -> rows of a support set of N rows: `pairs` = the ids m * H + h with idx[m, h] == j,
    grouped by j and ascending inside a group, `offsets` (N + 1).  Entries == N (the shadow row of the reference's padding,
    kpconv.py:89) and anything out of range are dropped.  Built once per table and frame with a stable device sort."""

    def __init__(self, idx: torch.Tensor, N: int):
        flat = idx.reshape(-1).to(torch.int32)     # 32-bit keys: half the radix passes of an int64 sort (13 tables per step)
        key, perm = torch.sort(flat, stable=True)
        bounds = torch.searchsorted(key, torch.arange(N + 1, device=idx.device, dtype=torch.int32))
        self.pairs = perm.to(torch.int32).contiguous()
        self.offsets = bounds.to(torch.int32).contiguous()   # rows j >= N lie behind offsets[N]: never visited
        self.N, self.M, self.H = N, idx.shape[0], (idx.shape[1] if idx.dim() == 2 else 1)


class TableCache:
    """Transposed tables of one frame's pyramid, keyed by (table address, column restriction): the same neighbour table serves the
    two or three KPConv layers of a stage and the max-pool of the next."""

    def __init__(self):
        self._t = {}

    def get(self, idx: torch.Tensor, N: int, first_column: bool = False) -> TransposedTable:
        # the entry keeps `idx` itself: a temporary index tensor (pix.to(int32), ...) would otherwise be freed after the forward op and the
        # allocator could hand its address to another index tensor of the same shape - which would silently get this table
        key = (idx.data_ptr(), tuple(idx.shape), N, first_column)
        ent = self._t.get(key)
        if ent is None:
            ent = self._t[key] = (idx, TransposedTable(idx[:, :1] if first_column and idx.dim() == 2 else idx, N))
        return ent[1]


# ------------------------------------------------------------------------------------------ helpers
def _pad4_cols(t: torch.Tensor) -> torch.Tensor:
    """(R, K) -> (R, roundup4(K)) zero padded: the GEMM contracts over multiples of 4 (16-byte operand loads)."""
    K = t.shape[1]
    if K % 4 == 0 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    out = torch.zeros((t.shape[0], (K + 3) // 4 * 4), dtype=t.dtype, device=t.device)
    out[:, :K].copy_(t)
    return out


def _rows(t: torch.Tensor) -> torch.Tensor:
    """A row-major view the kernels can read in place (unit column stride, 16-byte aligned rows - e.g. one part of a split (M, 3 C)
    projection, a column block of a concatenation's gradient), else a contiguous copy."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


def _gemm_nt(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a (M, K) . w (N, K)^T with operands padded to the GEMM's alignment rules."""
    return ops.gemm(_pad4_cols(a), _pad4_cols(w))


def col_sum(x: torch.Tensor) -> torch.Tensor:
    """out[c] = sum over the rows of x[:, c] (a bias gradient), fixed summation order."""
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.cofi_col_sum_workspace(M, C), dtype=torch.uint8, device=x.device)
    _lib.check(lib.cofi_col_sum(_p(x), _ld(x), M, C, _p(out), _p(ws), ws.numel(), _stream()), "cofi_col_sum")
    return out


# ------------------------------------------------------------------------------------------ Linear
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, rowdiv):
        xd, wd = x.detach(), w.detach()
        y = ops.gemm(_pad4_cols(xd), _pad4_cols(wd), bias=None if bias is None else bias.detach().contiguous(),
                     rowdiv=None if rowdiv is None else rowdiv.contiguous())
        ctx.save_for_backward(xd, wd, rowdiv)
        ctx.has_bias, ctx.arith = bias is not None, ops.gemm_mode()   # the backward (run later, by the caller's loss.backward()) computes in the same arithmetic
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, rowdiv = ctx.saved_tensors
        dy = _rows(dy)
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = col_sum(dy)                                                             # the bias is added after the division
        if rowdiv is not None:
            dy = dy / rowdiv[:, None]
        with ops.arithmetic(ctx.arith):
            if ctx.needs_input_grad[0]:
                dx = _gemm_nt(dy, ops.transpose(_rows(w)))[:, :x.shape[1]]         # dY W
            if ctx.needs_input_grad[1]:
                dw = _gemm_nt(*ops.transpose_pair(dy, _rows(x)))                         # dY^T X
        return dx, dw, db, None


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, rowdiv: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = (x w^T) / rowdiv[:, None] + bias on the MFMA GEMM, differentiable in x, w, bias."""
    return _Linear.apply(x, w, bias, rowdiv)


# ------------------------------------------------------------------------------------------ KPConv aggregation
class _KPConvAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, q_pts, s_pts, idx, kernel_points, sigma, table):
        f = feats.detach().contiguous()
        agg, cnt = ops.kpconv_aggregate(f, q_pts, s_pts, idx, kernel_points, sigma)
        ctx.save_for_backward(q_pts, s_pts, kernel_points)
        ctx.table, ctx.sigma, ctx.shape = table, float(sigma), f.shape
        ctx.mark_non_differentiable(cnt)
        ctx.set_materialize_grads(False)
        return agg, cnt

    @staticmethod
    def backward(ctx, dagg, _dcnt):
        if not ctx.needs_input_grad[0]:
            return (None,) * 7
        lib = _lib.load()
        q_pts, s_pts, kp = ctx.saved_tensors
        N, C = ctx.shape
        t = ctx.table
        dagg = _rows(dagg)
        df = torch.empty((N, C), dtype=torch.float32, device=dagg.device)
        rc = lib.cofi_kpconv_aggregate_bwd(_p(dagg), _ld(dagg), _p(q_pts), _p(s_pts), _p(t.pairs), _p(t.offsets), N, C, t.H, _p(kp), ctx.sigma,
                                           _p(df), _ld(df), _stream())
        _lib.check(rc, "cofi_kpconv_aggregate_bwd")
        return df, None, None, None, None, None, None


def kpconv_aggregate(feats, q_pts, s_pts, idx, kernel_points, sigma: float, tables: TableCache):
    """-> (agg (M, 15 C), cnt (M,)): kernel-point influences x neighbour features (kpconv.py:91-105) and the neighbour count the
    output is divided by (kpconv.py:113-116; not differentiable).  idx int32 (M, H) into the N rows of feats / s_pts."""
    table = tables.get(idx, feats.shape[0]) if feats.requires_grad else None
    return _KPConvAggregate.apply(feats, q_pts.contiguous(), s_pts.contiguous(), idx, kernel_points.detach().contiguous(), sigma, table)


# ------------------------------------------------------------------------------------------ neighbour max-pool / row gather
class _NeighborMaxpool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, table):
        lib = _lib.load()
        xd = x.detach().contiguous()
        N, C = xd.shape
        M, H = idx.shape
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((M, C), dtype=torch.uint8, device=x.device)
        _lib.check(lib.cofi_neighbor_maxpool_arg(_p(xd), _ld(xd), N, C, _p(idx), M, H, _p(out), _ld(out), _p(arg), _stream()), "cofi_neighbor_maxpool_arg")
        ctx.save_for_backward(arg)
        ctx.table, ctx.shape = table, (N, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        N, C = ctx.shape
        t = ctx.table
        dy = _rows(dy)
        dx = torch.empty((N, C), dtype=torch.float32, device=dy.device)
        _lib.check(lib.cofi_neighbor_maxpool_bwd(_p(dy), _ld(dy), _p(arg), C, t.H, _p(t.pairs), _p(t.offsets), N, _p(dx), _ld(dx), _stream()),
                   "cofi_neighbor_maxpool_bwd")
        return dx, None, None


def neighbor_maxpool(x, idx, tables: TableCache):
    """functional.py:53-66: max over the neighbours' rows (zero row behind idx == N)."""
    _mat(idx, "idx", torch.int32)
    return _NeighborMaxpool.apply(x, idx, tables.get(idx, x.shape[0]))


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, table):
        ctx.table, ctx.shape = table, x.shape
        return ops.gather_rows(x.detach().contiguous(), idx)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, C = ctx.shape
        t = ctx.table
        dy = _rows(dy)
        dx = torch.empty((N, C), dtype=torch.float32, device=dy.device)
        _lib.check(lib.cofi_gather_rows_bwd(_p(dy), _ld(dy), C, _p(t.pairs), _p(t.offsets), N, _p(dx), _ld(dx), _stream()), "cofi_gather_rows_bwd")
        return dx, None, None


def gather_rows(x, idx, tables: TableCache):
    """out[m] = x[idx[m, 0]] (zero row for idx == N): functional.py:5-21 nearest_upsample, or any row selection with idx (M,)."""
    return _GatherRows.apply(x, idx, tables.get(idx, x.shape[0], first_column=True))


# ------------------------------------------------------------------------------------------ convolution operand
class _Im2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W, ks, stride, pad):
        lib = _lib.load()
        xd = x.detach().contiguous()
        C = xd.shape[1]
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        col = torch.empty((Ho * Wo, ks * ks * C), dtype=torch.float32, device=x.device)
        _lib.check(lib.cofi_im2col_nhwc(_p(xd), _ld(xd), H, W, C, ks, stride, pad, _p(col), _ld(col), _stream()), "cofi_im2col_nhwc")
        ctx.geom = (H, W, C, ks, stride, pad)
        return col

    @staticmethod
    def backward(ctx, dcol):
        lib = _lib.load()
        H, W, C, ks, stride, pad = ctx.geom
        dcol = _rows(dcol)
        dx = torch.empty((H * W, C), dtype=torch.float32, device=dcol.device)
        _lib.check(lib.cofi_col2im_nhwc(_p(dcol), _ld(dcol), H, W, C, ks, stride, pad, _p(dx), _ld(dx), _stream()), "cofi_col2im_nhwc")
        return dx, None, None, None, None, None


def im2col(x, H: int, W: int, ks: int, stride: int = 1, pad: int = 1):
    """x (H W, C) pixel-major -> (Ho Wo, ks ks C), column (dy ks + dx) C + c (the order of image._nhwc_weight)."""
    if x.shape[0] != H * W or x.shape[1] % 4:
        raise _lib.CofiError("im2col: x must be (H*W, C) with C % 4 == 0")
    return _Im2col.apply(x, H, W, ks, stride, pad)


def conv2d(x, H: int, W: int, weight: torch.Tensor, stride: int = 1, pad: Optional[int] = None):
    """nn.Conv2d(bias=False) on a pixel-major map: weight (O, I, kh, kw) as the reference stores it.  -> (y (Ho Wo, O), Ho, Wo)."""
    ks = weight.shape[2]
    pad = ks // 2 if pad is None else pad
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    w2 = weight.permute(0, 2, 3, 1).reshape(weight.shape[0], -1)
    if ks == 1 and stride == 1:
        return linear(x, w2), Ho, Wo
    return linear(im2col(x, H, W, ks, stride, pad), w2), Ho, Wo


# ------------------------------------------------------------------------------------------ normalisation over rows + activation
class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, groups, slope, eps, fixed_stats):
        xd = x.detach().contiguous()
        stats = ops.group_stats(xd, groups, eps, exact=True) if fixed_stats is None else fixed_stats
        y = ops.group_norm_apply(xd, stats, None if gamma is None else gamma.detach().contiguous(), None if beta is None else beta.detach().contiguous(),
                                 slope=slope, res=None if res is None else res.detach().contiguous())
        ctx.save_for_backward(xd, y, stats, None if gamma is None else gamma.detach())
        ctx.cfg = (groups, float(slope), fixed_stats is not None, res is not None)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        lib = _lib.load()
        x, y, stats, gamma = ctx.saved_tensors
        groups, slope, const_stats, has_res = ctx.cfg
        M, C = x.shape
        dy = _rows(dy)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dg = torch.empty((C,), dtype=torch.float32, device=x.device) if gamma is not None else None
        db = torch.empty((C,), dtype=torch.float32, device=x.device) if gamma is not None else None
        ws = torch.empty(lib.cofi_group_norm_bwd_workspace(M, C, groups), dtype=torch.uint8, device=x.device)
        rc = lib.cofi_group_norm_bwd(_p(x), _ld(x), _p(y), _ld(y), _p(dy), _ld(dy), M, C, groups, _p(stats), _p(gamma), slope, int(const_stats),
                                     _p(dx), _ld(dx), _p(dg), _p(db), _p(dres), 0 if dres is None else _ld(dres), _p(ws), ws.numel(), _stream())
        _lib.check(rc, "cofi_group_norm_bwd")
        return dx, dg, db, dres, None, None, None, None


def group_norm_act(x, gamma=None, beta=None, groups: int = 32, slope: float = 1.0, res=None, eps: float = 1e-5, fixed_stats=None, return_stats: bool = False):
    """y = leaky(gn(x) * gamma + beta + res, slope): nn.GroupNorm(groups, C) over ALL rows of (rows, C) (modules.py:45-49) fused with the
    LeakyReLU / ReLU (slope 0.1 / 0) and the residual join behind it; groups == C, gamma = beta = None: the affine-less InstanceNorm of
    imagenet.py / network.py:42-43; groups == C with an affine pair: BatchNorm on batch statistics (fixed_stats (C, 2) = {mean, rstd}:
    on constant statistics).  HIP kernels forward (cofi_group_stats + cofi_group_norm_apply) and backward (cofi_group_norm_bwd)."""
    y, stats = _GroupNormAct.apply(x, gamma, beta, res, groups, slope, eps, fixed_stats)
    return (y, stats) if return_stats else y


# ------------------------------------------------------------------------------------------ attention
class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, nhead):
        qd, kd, vd = _rows(q.detach()), _rows(k.detach()), _rows(v.detach())
        o = ops.attention(qd, kd, vd, nhead=nhead)
        ctx.save_for_backward(qd, kd, vd, o)
        ctx.nhead = nhead
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        q, k, v, o = ctx.saved_tensors
        H = ctx.nhead
        L, HD = q.shape
        S, D = k.shape[0], HD // H
        do = _rows(do)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(lib.cofi_attention_bwd_workspace(L, H), dtype=torch.uint8, device=q.device)
        rc = lib.cofi_attention_bwd(_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o), _p(do), _ld(do), L, S, H, D, 1.0 / math.sqrt(D),
                                    _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv), _p(ws), ws.numel(), _stream())
        _lib.check(rc, "cofi_attention_bwd")
        return dq, dk, dv, None


class _Upsample2xCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, low, skip, h, w):
        ctx.geom = (h, w, low.shape[1])
        return ops.upsample2x_cat_nhwc(_rows(low.detach()), h, w, _rows(skip.detach()))

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        h, w, C1 = ctx.geom
        dout = _rows(dout)
        dlow = None
        if ctx.needs_input_grad[0]:
            dlow = torch.empty((h * w, C1), dtype=torch.float32, device=dout.device)
            _lib.check(lib.cofi_upsample2x_bwd_nhwc(_p(dout), _ld(dout), C1, h, w, _p(dlow), _ld(dlow), _stream()), "cofi_upsample2x_bwd_nhwc")
        return dlow, (dout[:, C1:] if ctx.needs_input_grad[1] else None), None, None


def upsample2x_cat(low: torch.Tensor, skip: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """imagenet.py:433-434: bilinear x2 (align_corners=False) of the pixel-major (h w, C1) map, concatenated with the (2h 2w, C2) skip map."""
    return _Upsample2xCat.apply(low, skip, h, w)


class _NormalizeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xd = _rows(x.detach())
        ctx.save_for_backward(xd)
        return ops.l2norm_rows(xd)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _rows(dy)
        M, C = x.shape
        dx = torch.empty((M, C), dtype=torch.float32, device=x.device)
        _lib.check(lib.cofi_l2norm_rows_bwd(_p(x), _ld(x), _p(dy), _ld(dy), M, C, 1e-12, _p(dx), C, _stream()), "cofi_l2norm_rows_bwd")
        return dx


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """F.normalize(x, p=2, dim=1) of a (rows, C) matrix (network.py:83-84, 90, 125-126: descriptors are unit rows)."""
    return _NormalizeRows.apply(x)


class _NormalizeCols(torch.autograd.Function):
    """y[:, c] = x[:, c] / max(||x[:, c]||_2, eps): F.normalize(x, dim=0), the token-axis normalisation of transformer.py:53, as
    `cofi_col_normalize` in both directions (column partials in a fixed order + apply).  torch's own reduction over the long axis splits
    a column over several workgroups behind a semaphore and did not survive hipGraph replays of the training step."""

    @staticmethod
    def _run(x, dy, stats, eps, bwd):
        lib = _lib.load()
        M, C = x.shape
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.cofi_col_normalize_workspace(M, C), dtype=torch.uint8, device=x.device)
        _lib.check(lib.cofi_col_normalize(_p(x), _ld(x), _p(dy), 0 if dy is None else _ld(dy), M, C, eps, bwd, _p(stats), _p(out), C, _p(ws), ws.numel(),
                                          _stream()), "cofi_col_normalize")
        return out

    @staticmethod
    def forward(ctx, x, eps):
        xd = x.detach()
        _mat(xd, "x")
        stats = torch.empty((2, xd.shape[1]), dtype=torch.float32, device=xd.device)
        y = _NormalizeCols._run(xd, None, stats, eps, 0)
        ctx.save_for_backward(xd, stats)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        dy = _rows(dy)
        return _NormalizeCols._run(x, dy, stats, ctx.eps, 1), None


def normalize_cols(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, p=2, dim=0) for a (rows, C) matrix."""
    return _NormalizeCols.apply(x, eps)


def attention(q, k, v, nhead: int = 4):
    """softmax(q k^T / sqrt(D)) v per head (linear_attention.py:56-79); q (L, H D), k / v (S, H D), D == 32."""
    return _Attention.apply(q, k, v, nhead)
