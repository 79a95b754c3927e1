"""The training losses of the reference (model/loss.py) on the GPU - SURVEY.md section 8 row f3, first part.

Same names, argument order and return values as the reference's functions (`device` first, kept for signature parity), each a
`torch.autograd.Function` over the HIP kernels of csrc/loss.hip: forward values AND the gradients w.r.t. the tensors the network
produced (descriptors, patches, scores) come from hand-written kernels; torch only carries the graph edges.  What is NOT here is the
backward of the network itself: these gradients end at the forward's outputs.

    loss_desc, dists = desc_loss(device, img_features_flatten_inline, pc_features_inline, correspondence_mask, pos_margin, neg_margin)   # train.py:254
    loss_coarse      = overlap_loss(device, coarse_pc_inline_score, coarse_pc_outline_score)                                            # train.py:260
    loss_fine        = fine_circle_loss(device, fine_img_feature_patch, fine_pc_inline_feature, relative_index, num_kpt)                # train.py:282
"""
import ctypes

import torch

from . import _lib
from .ops import _p, _stream


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.CofiError("%s must be a CUDA (HIP) tensor - there is no CPU path" % name)
    return t.detach().to(torch.float32).contiguous()


class _DescLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, pc, mask, pos_margin, neg_margin, log_scale):
        lib = _lib.load()
        img_c, pc_c, mask_c = _f32(img, "img_features"), _f32(pc, "pc_features"), _f32(mask, "mask")
        C, K = img_c.shape
        if pc_c.shape != (C, K) or mask_c.shape != (K, K):
            raise _lib.CofiError("desc_loss: img / pc features must be (C, K) and the mask (K, K) (loss.py:92 adds a per-row and a per-column term)")
        loss = torch.empty(1, dtype=torch.float32, device=img.device)
        dists = torch.empty((K, K), dtype=torch.float32, device=img.device)
        ws = torch.empty(lib.cofi_desc_loss_workspace(K), dtype=torch.uint8, device=img.device)
        _lib.check(lib.cofi_desc_loss(_p(img_c), K, _p(pc_c), K, _p(mask_c), C, K, pos_margin, neg_margin, log_scale, _p(loss), _p(dists),
                                      None, None, 0, None, 0, _p(ws), ws.numel(), _stream()), "cofi_desc_loss")
        ctx.save_for_backward(img_c, pc_c, mask_c)
        ctx.hp = (pos_margin, neg_margin, log_scale)
        ctx.mark_non_differentiable(dists)
        return loss.reshape(()), dists

    @staticmethod
    def backward(ctx, g_loss, _g_dists):
        lib = _lib.load()
        img_c, pc_c, mask_c = ctx.saved_tensors
        C, K = img_c.shape
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=img_c.device)
        dists = torch.empty((K, K), dtype=torch.float32, device=img_c.device)
        gi, gp = torch.empty_like(img_c), torch.empty_like(pc_c)
        ws = torch.empty(lib.cofi_desc_loss_workspace(K), dtype=torch.uint8, device=img_c.device)
        _lib.check(lib.cofi_desc_loss(_p(img_c), K, _p(pc_c), K, _p(mask_c), C, K, *ctx.hp, _p(loss), _p(dists), _p(g), _p(gi), K, _p(gp), K,
                                      _p(ws), ws.numel(), _stream()), "cofi_desc_loss")
        return gi, gp, None, None, None, None


def desc_loss(device, img_features, pc_features, mask, pos_margin=0.1, neg_margin=1.4, log_scale=10, num_kpt=512):
    """loss.py:69-93 -> (mean loss, dists).  img_features / pc_features (C, K) with unit columns, mask (K, K)."""
    return _DescLoss.apply(img_features, pc_features, mask, float(pos_margin), float(neg_margin), float(log_scale))


class _FineCircleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patches, pc, rel, m, gamma):
        lib = _lib.load()
        K = patches.shape[0]
        pt = _f32(patches, "fine_img_feature").reshape(K, -1, 16)
        C = pt.shape[1]
        pf = _f32(pc, "fine_pc_feature")
        if pf.shape != (K, C) or rel.numel() != K:
            raise _lib.CofiError("fine_circle_loss: patches (K, C, 4, 4), point descriptors (K, C), relative_index (K)")
        rel_c = rel.detach().to(torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pt.device)
        per = torch.empty(K, dtype=torch.float32, device=pt.device)
        _lib.check(lib.cofi_fine_circle_loss(_p(pt), _p(pf), C, _p(rel_c), K, C, m, gamma, _p(loss), _p(per), None, None, None, 0, _stream()),
                   "cofi_fine_circle_loss")
        ctx.save_for_backward(pt, pf, rel_c)
        ctx.hp, ctx.pshape = (m, gamma), patches.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_loss):
        lib = _lib.load()
        pt, pf, rel_c = ctx.saved_tensors
        K, C = pf.shape
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pt.device)
        per = torch.empty(K, dtype=torch.float32, device=pt.device)
        gpt, gpf = torch.empty_like(pt), torch.empty_like(pf)
        _lib.check(lib.cofi_fine_circle_loss(_p(pt), _p(pf), C, _p(rel_c), K, C, *ctx.hp, _p(loss), _p(per), _p(g), _p(gpt), _p(gpf), C, _stream()),
                   "cofi_fine_circle_loss")
        return gpt.reshape(ctx.pshape), gpf, None, None, None


def fine_circle_loss(device, fine_img_feature, fine_pc_feature, relative_index, num_kpt=64):
    """loss.py:9-51 (m = 0.2, gamma = 5).  fine_img_feature (K, C, 4, 4), fine_pc_feature (K, C), relative_index (K)."""
    return _FineCircleLoss.apply(fine_img_feature, fine_pc_feature, relative_index, 0.2, 5.0)


class _OverlapLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_in, s_out):
        lib = _lib.load()
        a, b = _f32(s_in, "inline_pc_score").reshape(-1), _f32(s_out, "outline_pc_score").reshape(-1)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.check(lib.cofi_overlap_loss(_p(a), a.numel(), _p(b), b.numel(), _p(loss), None, None, None, _stream()), "cofi_overlap_loss")
        ctx.save_for_backward(a, b)
        ctx.shapes = (s_in.shape, s_out.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_loss):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        _lib.check(lib.cofi_overlap_loss(_p(a), a.numel(), _p(b), b.numel(), None, _p(g), _p(ga), _p(gb), _stream()), "cofi_overlap_loss")
        return ga.reshape(ctx.shapes[0]), gb.reshape(ctx.shapes[1])


def overlap_loss(device, inline_pc_score, outline_pc_score):
    """loss.py:53-60: nn.BCELoss() of [inline | outline] scores against [1 | 0]."""
    return _OverlapLoss.apply(inline_pc_score, outline_pc_score)
