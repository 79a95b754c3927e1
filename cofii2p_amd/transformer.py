"""I2P transformer assembled from the HIP kernels (reference: model/transformer/transformer.py).

Token streams live in the LEFT half of (L, 2C) buffers so that the reference's
``torch.cat([x, message], dim=2)`` (transformer.py:61) is free: LayerNorm1 writes the message
straight into the right half and the 256-wide MLP GEMM reads the whole row.
"""
from typing import Dict

import torch

import os

from . import ops

JOINT_SELF = True   # the four self layers run ONCE over [image | point] tokens (shared weights); tests flip it for the per-stream form
# The fused layer tail owns 32 token rows per workgroup and streams ~1 MB of weight planes per tile through one wave per SIMD: built for
# the latency of a 1280-row call.  Above this many rows per call (stack-mode batches) the layer runs as plain GEMM launches instead,
# which are efficient at that size (batch 16, bf16x6: 165 us per fused tail vs ~110 us for merge + three GEMMs + the next projection;
# whole pipeline, frames/s with / without the fused tail: batch 16 601 / 619, batch 8 578 / 596, batch 4 555 / 569, batch 2 519 / 526,
# batch 1 - 1280 rows - 478 / 457).
TAIL_MAX_ROWS = int(os.environ.get("COFI_TAIL_MAX_ROWS", "2048"))


def pack_layer(sd: Dict[str, torch.Tensor], p: str) -> Dict[str, torch.Tensor]:
    """Fused projection weights of one LoFTREncoderLayer: [Wq;Wk;Wv] (3C,C) for self layers,
    [Wk;Wv] for cross layers; everything else is used in place (nn.Linear layout == GEMM layout).  The weights the fused layer tail
    reads (merge, MLP, and the stacked projections a PREVIOUS layer's tail computes for this one) also as bf16 planes:
    ".p2" = (2, N, K) hi / lo (3-term split), ".p3" = (3, N, K) hi / mid / lo (6-term split); ".f2" / ".f3" = the same planes in
    MFMA-fragment order (ops.fragment_order), the form the tail kernel streams."""
    wq, wk, wv = sd[p + "q_proj.weight"], sd[p + "k_proj.weight"], sd[p + "v_proj.weight"]
    out = {
        "q_proj.weight": wq.contiguous(),
        "kv.weight": torch.cat([wk, wv], 0).contiguous(),
        "qkv.weight": torch.cat([wq, wk, wv], 0).contiguous(),
        "merge.weight": sd[p + "merge.weight"].contiguous(),
        "mlp.0.weight": sd[p + "mlp.0.weight"].contiguous(),
        "mlp.2.weight": sd[p + "mlp.2.weight"].contiguous(),
        "norm1.weight": sd[p + "norm1.weight"].contiguous(), "norm1.bias": sd[p + "norm1.bias"].contiguous(),
        "norm2.weight": sd[p + "norm2.weight"].contiguous(), "norm2.bias": sd[p + "norm2.bias"].contiguous(),
    }
    for name in ("merge", "mlp.0", "mlp.2", "qkv", "kv"):
        for n in (2, 3):
            out["%s.p%d" % (name, n)] = ops.split_planes(out[name + ".weight"], n)
            out["%s.f%d" % (name, n)] = ops.fragment_order(out["%s.p%d" % (name, n)])   # ... and in fragment order (ops.TAIL_FRAG)
    return out


def _layer(w, xcat: torch.Tensor, src: torch.Tensor, out: torch.Tensor, self_attn: bool, nhead: int = 4, frames: int = 1):
    """xcat (L,2C): x in [:, :C]; src (S,C) view; out (L,C) view receiving x + message.
    transformer.py:43-64."""
    C = xcat.shape[1] // 2
    x = xcat[:, :C]
    # projections; the token-axis norm of Q (F.normalize over dim=1, transformer.py:53) comes from the GEMM's
    # fused column statistics
    if self_attn:
        qkv, part = ops.gemm_colstats(x, w["qkv.weight"])
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        q, part = ops.gemm_colstats(x, w["q_proj.weight"])
        kv = ops.gemm(src, w["kv.weight"])
        k, v = kv[:, :C], kv[:, C:]
    # the attention kernel folds the partials into the per-frame token-axis norm of Q itself
    fused_tail = ops.tail_planes() > 0 and C == 128 and x.shape[0] <= TAIL_MAX_ROWS
    rows_q = x.shape[0] // frames
    if frames > 1 and rows_q % 64:   # the projection's 64-row statistics slabs straddle frames: explicit per-frame column norms
        qs = torch.stack([ops.col_inv_norm(q[f * rows_q:(f + 1) * rows_q]) for f in range(frames)])
        msg = ops.attention(q, k, v, q_colscale=qs, nhead=nhead, frames=frames, parts=fused_tail)
    else:
        msg = ops.attention(q, k, v, q_colpart=part, nhead=nhead, frames=frames, parts=fused_tail)
    if fused_tail:
        # merge + LN1 + concat + MLP + LN2 + residual: one kernel, intermediates stay in LDS; its loader also combines the
        # attention kernel's partial slots
        return ops.loftr_tail(msg, x, w, out)
    # merge Linear + LayerNorm1 in one kernel, written into the right half of the concat buffer
    ops.gemm_layernorm(msg, w["merge.weight"], w["norm1.weight"], w["norm1.bias"], out=xcat[:, C:])
    h = ops.gemm(xcat, w["mlp.0.weight"], act=ops.ACT_RELU)
    # second MLP Linear + LayerNorm2 + residual in one kernel
    ops.gemm_layernorm(h, w["mlp.2.weight"], w["norm2.weight"], w["norm2.bias"], res=x, out=out)
    return out


def loftr_layer(w, x: torch.Tensor, src: torch.Tensor, nhead: int = 4) -> torch.Tensor:
    """Stand-alone layer on plain (L,C)/(S,C) tensors (tests); `w` holds raw reference-named weights
    or a packed dict."""
    if "kv.weight" not in w:
        w = pack_layer(w, "")
    L, C = x.shape
    xcat = torch.empty((L, 2 * C), dtype=torch.float32, device=x.device)
    xcat[:, :C].copy_(x)
    out = torch.empty((L, C), dtype=torch.float32, device=x.device)
    return _layer(w, xcat, src, out, self_attn=False, nhead=nhead)


class TokenStreams:
    """Ping-pong (L,2C) buffers for the image and point streams.  Both streams of one ping-pong phase share ONE
    allocation ([image rows | point rows]): a self layer, whose weights are shared by the two streams, then runs
    over both at once (stack mode with twice the frames) when the streams have equally many tokens per frame."""

    def __init__(self, L_img: int, L_pc: int, C: int, device):
        self.C = C
        self.both = [torch.empty((L_img + L_pc, 2 * C), dtype=torch.float32, device=device) for _ in range(2)]
        self.img = [b[:L_img] for b in self.both]
        self.pc = [b[L_img:] for b in self.both]
        self.joint = L_img == L_pc
        self.cur_img = 0
        self.cur_pc = 0

    def img_tokens(self):
        return self.img[self.cur_img][:, : self.C]

    def pc_tokens(self):
        return self.pc[self.cur_pc][:, : self.C]


FUSED_CHAIN = os.environ.get("COFI_TRANSFORMER_CHAIN", "1") != "0"


def _chain_ok(ts: "TokenStreams", kinds, frames: int) -> bool:
    """The fused chain (every tail also computes the projections of the layers that follow) serves the reference's configuration:
    d_model 128, alternating self / cross layers starting with a self layer, equally many image and point tokens per frame in
    multiples of 64 (the first projection's statistics slabs; the tails work on 32-row tiles - KITTI: 1280 / 1280), one of the bf16-split arithmetics."""
    L = ts.img[0].shape[0] // frames
    return (FUSED_CHAIN and JOINT_SELF and ts.C == 128 and ops.tail_planes() > 0 and ts.joint and L % 64 == 0 and len(kinds) >= 2
            and ts.img[0].shape[0] <= TAIL_MAX_ROWS
            and all(k == ("self", "cross")[i & 1] for i, k in enumerate(kinds)))


def _run_chain(layers, kinds, ts: "TokenStreams", nhead: int, frames: int, l2=None):
    """transformer.py:85-104 as 1 + 3 launches per layer pair and direction less than the layer-by-layer form: ONE projection GEMM in
    front of layer 0, then per self layer attention + tail, per cross layer (attention + tail) x 2 - every tail computes, from its
    32-row tile of `out` still in LDS, the q / k / v the following attention calls read (and the column partials of q for the token-axis
    norm).  l2 = (img_l2, pc_l2, img_l2t, pc_l2t): F.normalize(dim=1) of the last layer's outputs written by its tails (entries may be None)."""
    C, sfx = ts.C, ops.tail_suffix()
    T = ts.img[0].shape[0]            # rows per stream (frames * tokens)
    dev = ts.both[0].device
    qkv = [torch.empty((2 * T, 3 * C), dtype=torch.float32, device=dev) for _ in range(2)]
    part = [torch.empty((2 * T // 32, 3 * C, 2), dtype=torch.float32, device=dev) for _ in range(2)]
    kv2 = torch.empty((T, 2 * C), dtype=torch.float32, device=dev)
    ns = T // 32                      # 32-row slabs per stream
    cur = 0
    # layer 0 (self): the only stand-alone projection
    xb = ts.both[ts.cur_img]
    _, part0 = ops.gemm_colstats(xb[:, :C], layers[0]["qkv.weight"], out=qkv[cur])
    cur_part = part0                  # 64-row slabs from the GEMM epilogue; the tails write 32-row slabs into part[...]
    nl = len(kinds)
    for li, (w, kind) in enumerate(zip(layers, kinds)):
        nxt = layers[li + 1] if li + 1 < nl else None
        q_all = qkv[cur]
        if kind == "self":
            xb, ob = ts.both[ts.cur_img], ts.both[ts.cur_img ^ 1]
            msg = ops.attention(q_all[:, :C], q_all[:, C:2 * C], q_all[:, 2 * C:], q_colpart=cur_part, nhead=nhead, frames=2 * frames, parts=True)
            proj = [] if nxt is None else [(nxt["qkv" + sfx], qkv[cur ^ 1], part[cur ^ 1])]
            ops.loftr_tail(msg, xb[:, :C], w, ob[:, :C], proj=proj)
            cur, cur_part = cur ^ 1, part[cur ^ 1]
        else:
            xi, xp = ts.img[ts.cur_img], ts.pc[ts.cur_pc]
            oi, op = ts.img[ts.cur_img ^ 1], ts.pc[ts.cur_pc ^ 1]
            half = cur_part.shape[0] // 2
            last = nxt is None
            # image tokens attend to the point tokens (transformer.py:99): q of the image rows, k / v of the point rows
            msg = ops.attention(q_all[:T, :C], q_all[T:, C:2 * C], q_all[T:, 2 * C:], q_colpart=cur_part[:half], nhead=nhead, frames=frames, parts=True)
            proj = [(w["kv" + sfx], kv2, None)]
            if not last:
                proj.append((nxt["qkv" + sfx], qkv[cur ^ 1][:T], part[cur ^ 1][:ns]))
            ops.loftr_tail(msg, xi[:, :C], w, oi[:, :C], proj=proj, out_l2=l2[0] if (last and l2) else None, out_l2t=l2[2] if (last and l2) else None)
            # point tokens attend to the UPDATED image tokens (transformer.py:100): k / v from the image tail's projection
            msg = ops.attention(q_all[T:, :C], kv2[:, :C], kv2[:, C:], q_colpart=cur_part[half:], nhead=nhead, frames=frames, parts=True)
            proj = [] if last else [(nxt["qkv" + sfx], qkv[cur ^ 1][T:], part[cur ^ 1][ns:])]
            ops.loftr_tail(msg, xp[:, :C], w, op[:, :C], proj=proj, out_l2=l2[1] if (last and l2) else None, out_l2t=l2[3] if (last and l2) else None)
            cur, cur_part = cur ^ 1, part[cur ^ 1]
        ts.cur_img ^= 1
        ts.cur_pc ^= 1
    return ts.img_tokens(), ts.pc_tokens()


def run_transformer(layers, kinds, ts: TokenStreams, nhead: int = 4, frames: int = 1, l2=None):
    """transformer.py:85-104.  Self layers share weights between the streams; in a cross layer the
    point stream attends to the ALREADY UPDATED image stream (:99-100).  l2 (optional, the fused chain only - see _run_chain):
    destinations for the L2-normalised outputs; returns (img tokens, pc tokens, l2 written?)."""
    C = ts.C
    if _chain_ok(ts, kinds, frames):
        ti, tp = _run_chain(layers, kinds, ts, nhead, frames, l2 if kinds[-1] == "cross" else None)
        return ti, tp, (l2 is not None and kinds[-1] == "cross")
    for w, kind in zip(layers, kinds):
        xi, xp = ts.img[ts.cur_img], ts.pc[ts.cur_pc]
        oi, op = ts.img[ts.cur_img ^ 1], ts.pc[ts.cur_pc ^ 1]
        if kind == "self" and ts.joint and ts.cur_img == ts.cur_pc and JOINT_SELF:
            # shared weights, independent streams: one pass over [image tokens | point tokens] as 2*frames frames
            xb, ob = ts.both[ts.cur_img], ts.both[ts.cur_img ^ 1]
            _layer(w, xb, xb[:, :C], ob[:, :C], True, nhead, 2 * frames)
        elif kind == "self":  # the two modalities are independent here: point stream on a side HIP stream
            with ops.Branch(xi.device, 2) as br:
                _layer(w, xp, xp[:, :C], op[:, :C], True, nhead, frames)
            _layer(w, xi, xi[:, :C], oi[:, :C], True, nhead, frames)
            br.join(op)
        else:
            _layer(w, xi, xp[:, :C], oi[:, :C], False, nhead, frames)
            _layer(w, xp, oi[:, :C], op[:, :C], False, nhead, frames)
        ts.cur_img ^= 1
        ts.cur_pc ^= 1
    return ts.img_tokens(), ts.pc_tokens(), False
