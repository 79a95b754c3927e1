"""Row f3 of SURVEY.md section 8: the training path.  (a) every autograd Function of cofii2p_amd/autograd.py - forward value and
gradients - against the same operator written with plain torch ops and differentiated by torch.autograd; (b) one optimisation step
shaped like train.py:186-285 on the tiny frame against tests/golden/train_ref.npz, recorded from the REFERENCE's model in train() mode
(tests/tools/make_golden_train.py): train-mode outputs, the three losses, the gradient of every parameter, the BatchNorm running
statistics.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from common import frame_inputs, load_golden, sha  # noqa: E402

DEV = "cuda:0"
GRAD_TOL = 1e-3   # VERDICT r2 / north star: parameter gradients within 1e-3 relative of the reference's


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if isinstance(a, np.ndarray) else a.to(DEV)
    return t.requires_grad_() if grad else t


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def f32_arithmetic(monkeypatch):
    from cofii2p_amd import ops

    monkeypatch.setattr(ops, "GEMM_MODE", "f32")


# ------------------------------------------------------------------------------------------ (a) operators
@pytest.mark.parametrize("M,K,N,bias,rowdiv", [(200, 96, 64, True, False), (130, 15 * 32, 32, True, True), (77, 64, 1, False, False), (50, 6, 10, True, False)])
def test_linear_function(M, K, N, bias, rowdiv):
    from cofii2p_amd import autograd as ag

    g = torch.Generator(device=DEV).manual_seed(M)
    x, w = torch.randn((M, K), device=DEV, generator=g), torch.randn((N, K), device=DEV, generator=g) / math.sqrt(K)
    b = torch.randn((N,), device=DEV, generator=g) if bias else None
    rd = torch.randint(1, 9, (M,), device=DEV, generator=g).float() if rowdiv else None
    dy = torch.randn((M, N), device=DEV, generator=g)
    ins = [t.clone().requires_grad_() for t in (x, w)] + ([b.clone().requires_grad_()] if bias else [])
    y = ag.linear(ins[0], ins[1], ins[2] if bias else None, rd)
    y.backward(dy)
    ref_in = [t.double().clone().requires_grad_() for t in (x, w)] + ([b.double().clone().requires_grad_()] if bias else [])
    yr = ref_in[0] @ ref_in[1].t()
    if rowdiv:
        yr = yr / rd.double()[:, None]
    if bias:
        yr = yr + ref_in[2]
    yr.backward(dy.double())
    assert rel_err(y, yr) < 1e-5
    for a, r in zip(ins, ref_in):
        assert rel_err(a.grad, r.grad) < 1e-5


def _kpconv_agg_torch(feats, q_pts, s_pts, idx, kp, sigma):
    """kpconv.py:89-105 with plain torch ops (differentiable in feats)."""
    N = feats.shape[0]
    sp = torch.cat([s_pts, torch.full((1, 3), 1e6, device=feats.device, dtype=s_pts.dtype)], 0)
    sf = torch.cat([feats, torch.zeros((1, feats.shape[1]), device=feats.device, dtype=feats.dtype)], 0)
    ii = idx.long()
    rel = sp[ii] - q_pts[:, None, :]
    infl = (1.0 - (rel[:, :, None, :] - kp[None, None]).pow(2).sum(-1).sqrt() / sigma).clamp_min(0.0)   # (M, H, K)
    nf = sf[ii]
    agg = infl.transpose(1, 2) @ nf                                                                        # (M, K, C)
    return agg.reshape(agg.shape[0], -1)


@pytest.mark.parametrize("C", [32, 64, 128, 16])
def test_kpconv_aggregate_function(C):
    from cofii2p_amd import autograd as ag
    from cofii2p_amd.weights import kernel_point_table

    g = np.random.default_rng(C)
    N, M, H, sigma = 300, 180, 24, 0.3
    s_pts = G((g.random((N, 3)) * 1.2).astype(np.float32))
    q_pts = G((s_pts[:M].cpu().numpy() + g.normal(0, 0.02, (M, 3))).astype(np.float32))
    idx_np = g.integers(0, N, (M, H))
    idx_np[::9, -2:] = N        # shadow neighbours (kpconv.py:89)
    idx_np[3, :] = N
    idx_np[5, :4] = idx_np[5, 4]   # one support row several times in a neighbourhood (duplicates of the sub-sampling with replacement)
    idx = G(idx_np.astype(np.int32))
    kp = G(kernel_point_table(15, 0.425 * sigma / 0.2))
    feats = G(g.standard_normal((N, C)).astype(np.float32))
    dagg = G(g.standard_normal((M, 15 * C)).astype(np.float32))
    f1 = feats.clone().requires_grad_()
    agg, cnt = ag.kpconv_aggregate(f1, q_pts, s_pts, idx, kp, sigma, ag.TableCache())
    agg.backward(dagg)
    f2 = feats.double().clone().requires_grad_()
    ref = _kpconv_agg_torch(f2, q_pts.double(), s_pts.double(), idx, kp.double(), sigma)
    ref.backward(dagg.double())
    assert rel_err(agg, ref) < 1e-5
    assert rel_err(f1.grad, f2.grad) < 1e-5
    # bit-reproducible: the transposed-table walk has a fixed summation order
    f3 = feats.clone().requires_grad_()
    ag.kpconv_aggregate(f3, q_pts, s_pts, idx, kp, sigma, ag.TableCache())[0].backward(dagg)
    assert torch.equal(f1.grad, f3.grad)


def test_neighbor_maxpool_and_gather_functions():
    from cofii2p_amd import autograd as ag

    g = np.random.default_rng(3)
    N, M, H, C = 150, 90, 16, 96
    x = G(g.standard_normal((N, C)).astype(np.float32))
    x[10] = x[11]                      # duplicate rows: a tie, one of them takes the gradient
    idx_np = g.integers(0, N, (M, H))
    idx_np[::7, -3:] = N
    idx_np[4, :] = N                   # only the zero pad row: output 0, no gradient anywhere
    idx_np[6, 0], idx_np[6, 1] = 10, 11
    idx = G(idx_np.astype(np.int32))
    dy = G(g.standard_normal((M, C)).astype(np.float32))
    x1 = x.clone().requires_grad_()
    tables = ag.TableCache()
    y = ag.neighbor_maxpool(x1, idx, tables)
    y.backward(dy)
    x2 = x.clone().requires_grad_()
    xp = torch.cat([x2, torch.zeros((1, C), device=DEV)], 0)
    yr = xp[idx.long()].max(1)[0]
    yr.backward(dy)
    assert torch.equal(y, yr)
    # ties may send the gradient to another of the equal rows: compare with the tied rows folded together
    ga, gb = x1.grad.clone(), x2.grad.clone()
    ga[10] += ga[11]; gb[10] += gb[11]; ga[11] = 0; gb[11] = 0
    assert rel_err(ga, gb) < 1e-6
    # nearest up-sample: column 0 of the table (functional.py:20)
    x3, x4 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y = ag.gather_rows(x3, idx, tables)
    y.backward(dy)
    yr = torch.cat([x4, torch.zeros((1, C), device=DEV)], 0)[idx[:, 0].long()]
    yr.backward(dy)
    assert torch.equal(y, yr) and rel_err(x3.grad, x4.grad) < 1e-6
    # 1-D index list with repeats (the fine key-point selection, network.py:137)
    sel = G(g.integers(0, N, 40).astype(np.int32))
    x5, x6 = x.clone().requires_grad_(), x.clone().requires_grad_()
    ag.gather_rows(x5, sel, ag.TableCache()).backward(dy[:40])
    x6[sel.long()].backward(dy[:40])
    assert rel_err(x5.grad, x6.grad) < 1e-6


@pytest.mark.parametrize("H,W,Cin,Cout,ks,stride,pad", [(12, 20, 8, 16, 3, 1, 1), (13, 21, 8, 12, 3, 2, 1), (10, 16, 16, 8, 1, 2, 0), (9, 11, 4, 8, 1, 1, 0)])
def test_conv2d_function(H, W, Cin, Cout, ks, stride, pad):
    from cofii2p_amd import autograd as ag

    g = torch.Generator(device=DEV).manual_seed(H * W)
    x = torch.randn((1, Cin, H, W), device=DEV, generator=g)
    w = torch.randn((Cout, Cin, ks, ks), device=DEV, generator=g) / math.sqrt(Cin * ks * ks)
    x1, w1 = x.reshape(Cin, -1).t().contiguous().requires_grad_(), w.clone().requires_grad_()
    y, Ho, Wo = ag.conv2d(x1, H, W, w1, stride, pad)
    x2, w2 = x.double().clone().requires_grad_(), w.double().clone().requires_grad_()
    yr = F.conv2d(x2, w2, None, stride, pad)
    assert yr.shape[2:] == (Ho, Wo)
    dy = torch.randn((Ho * Wo, Cout), device=DEV, generator=g)
    y.backward(dy)
    yr.backward(dy.t().reshape(1, Cout, Ho, Wo).double())
    assert rel_err(y, yr[0].reshape(Cout, -1).t()) < 1e-5
    assert rel_err(w1.grad, w2.grad) < 1e-5
    assert rel_err(x1.grad, x2.grad[0].reshape(Cin, -1).t()) < 1e-5


@pytest.mark.parametrize("L,S", [(64, 96), (70, 45), (1280, 128), (33, 1280)])
def test_attention_function(L, S):
    from cofii2p_amd import autograd as ag

    H, D = 4, 32
    g = torch.Generator(device=DEV).manual_seed(L + S)
    q = torch.randn((L, H * D), device=DEV, generator=g) * 0.7
    k, v = torch.randn((S, H * D), device=DEV, generator=g), torch.randn((S, H * D), device=DEV, generator=g)
    do = torch.randn((L, H * D), device=DEV, generator=g)
    a = [t.clone().requires_grad_() for t in (q, k, v)]
    o = ag.attention(*a, nhead=H)
    o.backward(do)
    r = [t.double().clone().requires_grad_() for t in (q, k, v)]
    qh, kh, vh = (t.reshape(-1, H, D) for t in r)
    A = torch.softmax(torch.einsum("lhd,shd->hls", qh, kh) / math.sqrt(D), dim=-1)    # linear_attention.py:70-76
    orf = torch.einsum("hls,shd->lhd", A, vh).reshape(L, H * D)
    orf.backward(do.double())
    assert rel_err(o, orf) < 1e-5
    for x, y, n in zip(a, r, "qkv"):
        assert rel_err(x.grad, y.grad) < 2e-5, n


@pytest.mark.parametrize("M,C", [(20480, 64), (1280, 128), (7, 128), (33, 20)])
def test_normalize_rows_function(M, C):
    """F.normalize(x, dim=1), forward and backward, incl. an all-zero row (clamped norm: gradient dy / eps = 0 for dy = 0 there)."""
    from cofii2p_amd import autograd as ag

    g = torch.Generator().manual_seed(M + C)
    base, dy = torch.randn((M, C), generator=g), torch.randn((M, C), generator=g)
    base[3] = 0.0
    dy[3] = 0.0
    xr = base.clone().double().requires_grad_()
    yr = F.normalize(xr, dim=1)
    yr.backward(dy.double())
    x = G(base, grad=True)
    y = ag.normalize_rows(x)
    y.backward(G(dy))
    assert rel_err(y, yr) < 1e-6 and rel_err(x.grad, xr.grad) < 2e-6


@pytest.mark.parametrize("h,w,C1,C2", [(20, 64, 128, 128), (5, 7, 8, 4), (1, 3, 4, 8), (8, 1, 4, 4)])
def test_upsample2x_cat_function(h, w, C1, C2):
    """imagenet.py:433-434: bilinear x2 (align_corners=False) + concatenation, forward and both gradients (the adjoint gathers per input pixel)."""
    from cofii2p_amd import autograd as ag

    g = torch.Generator().manual_seed(h * 100 + w)
    low, skip, dy = torch.randn((h * w, C1), generator=g), torch.randn((4 * h * w, C2), generator=g), torch.randn((4 * h * w, C1 + C2), generator=g)
    lr, sr = low.clone().double().requires_grad_(), skip.clone().double().requires_grad_()
    up = F.interpolate(lr.t().reshape(1, C1, h, w), scale_factor=2, mode="bilinear", align_corners=False)
    yr = torch.cat([up.reshape(C1, 4 * h * w).t(), sr], 1)
    yr.backward(dy.double())
    l, s_ = G(low, grad=True), G(skip, grad=True)
    y = ag.upsample2x_cat(l, s_, h, w)
    y.backward(G(dy))
    assert rel_err(y, yr) < 1e-6 and rel_err(l.grad, lr.grad) < 1e-6 and torch.equal(s_.grad.cpu(), dy[:, C1:])


@pytest.mark.parametrize("M,C1,C2", [(20480, 64, 576), (130, 6, 33), (64, 64, 64), (1, 8, 4)])
def test_transpose_pair(M, C1, C2):
    from cofii2p_amd import ops

    g = torch.Generator().manual_seed(M)
    a, b = G(torch.randn((M, C1 + 4), generator=g))[:, :C1], G(torch.randn((M, C2), generator=g))
    at, bt = ops.transpose_pair(a, b)
    assert torch.equal(at, a.t().contiguous()) and torch.equal(bt, b.t().contiguous())


@pytest.mark.parametrize("L,C,view", [(1280, 128, True), (96, 128, False), (8, 128, False), (333, 64, True)])
def test_normalize_cols_function(L, C, view):
    """F.normalize(x, dim=0) (transformer.py:53 normalises Q over the tokens), forward and backward, incl. a dead (all-zero) column."""
    from cofii2p_amd import autograd as ag

    g = torch.Generator().manual_seed(L + C)
    base = torch.randn((L, 3 * C if view else C), generator=g)
    base[:, 5] = 0.0
    dy = torch.randn((L, C), generator=g)
    xr = base.clone().double().requires_grad_()
    yr = F.normalize(xr[:, :C], dim=0)
    yr.backward(dy.double())
    x = G(base, grad=True)
    y = ag.normalize_cols(x[:, :C])
    y.backward(G(dy))
    assert rel_err(y, yr) < 2e-6 and rel_err(x.grad, xr.grad) < 5e-6
    assert float(y.detach()[:, 5].abs().max()) == 0.0


@pytest.mark.parametrize("M,C,groups,affine,slope,res,fixed", [
    (2048, 64, 32, True, 0.1, False, False),     # UnaryBlock / ConvBlock: GroupNorm + LeakyReLU
    (1000, 256, 32, True, 0.1, True, False),     # residual tail: leaky(GroupNorm(unary2) + shortcut), ragged row count
    (333, 2048, 32, True, 1.0, False, False),    # shortcut branch: no activation, 64 channels per group
    (5120, 64, 64, False, 0.0, True, False),     # BasicBlock end: relu(InstanceNorm(conv2) + identity)
    (1280, 128, 128, False, 0.0, False, False),  # score head: InstanceNorm + ReLU
    (10240, 128, 128, True, 0.0, True, False),   # ResidualConv end, train(): relu(BatchNorm(batch statistics) + identity)
    (640, 64, 64, True, 0.0, False, True),       # BatchNorm on constant (running) statistics
    (256, 128, 32, True, 0.1, False, False), (128, 512, 32, True, 0.1, True, False), (256, 512, 32, True, 0.1, True, False),   # deep stages of the tiny frame
    (64, 32, 32, True, 0.1, False, False),
])
def test_group_norm_act_function(M, C, groups, affine, slope, res, fixed):
    from cofii2p_amd import autograd as ag

    g = torch.Generator(device=DEV).manual_seed(M + C)
    x = torch.randn((M, C), device=DEV, generator=g) * 1.7 + 0.3
    ga = (1 + 0.2 * torch.randn((C,), device=DEV, generator=g)) if affine else None
    be = (0.3 * torch.randn((C,), device=DEV, generator=g)) if affine else None
    r = torch.randn((M, C), device=DEV, generator=g) if res else None
    dy = torch.randn((M, C), device=DEV, generator=g)
    fs = None
    if fixed:
        rm, rv = torch.randn((C,), device=DEV, generator=g) * 0.2, torch.rand((C,), device=DEV, generator=g) + 0.5
        fs = torch.stack([rm, torch.rsqrt(rv + 1e-5)], 1).contiguous()
    ins = [t.clone().requires_grad_() if t is not None else None for t in (x, ga, be, r)]
    y = ag.group_norm_act(ins[0], ins[1], ins[2], groups, slope, ins[3], fixed_stats=fs)
    y.backward(dy)
    ref = [t.double().clone().requires_grad_() if t is not None else None for t in (x, ga, be, r)]
    if fixed:
        z = (ref[0] - rm.double()) * torch.rsqrt(rv.double() + 1e-5) * ref[1] + ref[2]
    else:
        z = F.group_norm(ref[0].t().unsqueeze(0), groups, ref[1], ref[2], 1e-5).squeeze(0).t()
    if res:
        z = z + ref[3]
    yr = F.leaky_relu(z, slope) if slope != 1.0 else z
    yr.backward(dy.double())
    assert rel_err(y, yr) < 1e-5
    for a, b, n in zip(ins, ref, ("x", "gamma", "beta", "res")):
        if a is not None:
            assert rel_err(a.grad, b.grad) < 2e-5, n


# ------------------------------------------------------------------------------------------ (b) one step of train.py vs the reference
class Opt:
    img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"


@pytest.fixture(scope="module")
def gold():
    return load_golden("train_ref.npz")


def _check_step_against(gold, m, tol, zero_mult=3.0, err32_mult=2.5, err32_floor=0.0, ratios=None):
    """one train.py-shaped step of module `m` against a train_ref*.npz recording: outputs, mask, losses, gradients (judged in float64, see
    test_train_step_matches_reference), buffers.  -> (worst error, its parameter)"""
    from cofii2p_amd.train_step import step_losses

    dd, img, batch, sopt = _train_inputs(gold)
    outs, mask, (l_desc, l_coarse, l_fine) = step_losses(m, dd, img, batch, sopt)
    assert np.array_equal(mask.cpu().numpy(), gold["mask"])
    for n_, t in zip(("img_desc", "pc_desc", "img_score", "pc_score", "patches", "fine_pc"), outs[:6]):
        ref = gold["train_" + n_]
        assert tuple(t.shape) == ref.shape, n_
        assert float((t.detach().cpu() - torch.from_numpy(ref)).abs().max()) < 1e-3, n_
    for name, val in (("loss_desc", l_desc), ("loss_coarse", l_coarse), ("loss_fine", l_fine)):
        assert abs(float(val.detach()) - float(gold[name])) < 1e-3 * max(1.0, abs(float(gold[name]))), name
    (l_desc + l_coarse + l_fine).backward()
    names = [str(n) for n in gold["g_names"]]
    params = dict(m.named_parameters())
    assert names == list(params)
    total = float(np.sqrt((gold["g_norm64"] ** 2).sum()))
    worst = (0.0, None)
    bad = []
    if err32_floor == "median":   # the reference's own typical fp32 deviation in this configuration (1.7e-6 'gn' / 'ln', 1.8e-3 'bn')
        live = (gold["g_has"] != 0) & (gold["g_norm64"] >= 1e-6 * total)
        err32_floor = float(np.median(gold["g_err32"][live]))
    for i, name in enumerate(names):
        p = params[name]
        if not int(gold["g_has"][i]):
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        flat = p.grad.detach().double().reshape(-1).cpu()
        norm_ref = float(gold["g_norm64"][i])
        if norm_ref < 1e-6 * total:
            assert float(flat.norm()) < max(1e-5 * total, zero_mult * float(gold["g_norm"][i])), "%s: mathematically zero gradient, got |g| = %.3g" % (name, float(flat.norm()))
            continue
        got, ref = flat[torch.from_numpy(gold["g_pos"][i])].numpy(), gold["g_val64"][i]
        scale = max(np.linalg.norm(ref), norm_ref * math.sqrt(len(ref) / flat.numel()))
        e_samp = float(np.linalg.norm(got - ref) / scale)
        e_norm = abs(float(flat.norm()) - norm_ref) / norm_ref
        allow = max(tol, err32_mult * max(float(gold["g_err32"][i]), err32_floor))
        worst = max(worst, (max(e_samp, e_norm) / allow, name))
        if ratios is not None:   # statistical judgement by the caller: error / allowance of every parameter
            ratios.append((max(e_samp, e_norm) / allow, name))
            continue
        if not (e_samp < allow and e_norm < allow):
            bad.append("%s: sampled entries off by %.3g (allowed %.3g; reference fp32 %.3g), norm by %.3g (relative)" % (name, e_samp, allow, float(gold["g_err32"][i]), e_norm))
    assert not bad, "%d parameters beyond their allowance:\n" % len(bad) + "\n".join(bad[:40])
    bufs = dict(m.named_buffers())
    nbuf = 0
    for k in gold.files:
        if k.startswith("buf/"):
            nbuf += 1
            # BatchNorm1d of the point encoder ('bn'): the batch statistics inherit the forward's fp32 conditioning - 1e-4 relative at the last
            # stage between any two fp32 evaluations (tools/diag_norm_train.py) - times the momentum 0.1
            rtol, atol = (2e-3, 2e-4) if k.startswith("buf/pc_encoder") else (1e-4, 1e-5)
            got, ref = bufs[k[4:]].detach().cpu().double(), torch.from_numpy(np.asarray(gold[k])).double()
            assert torch.allclose(got, ref, rtol=rtol, atol=atol), "%s: max abs difference %.3g" % (k, float((got - ref).abs().max()))
    return worst, nbuf


@pytest.mark.parametrize("norm", ["bn", "ln"])
def test_train_step_other_point_norms_match_reference(norm):
    """opt.norm = 'bn' / 'ln' (get_norm(), modules.py:51-60) in train() mode against the REFERENCE built with that option and run through one
    train.py-shaped step (tests/golden/train_ref_{bn,ln}.npz, tests/tools/make_golden_train.py --norm ...): train-mode outputs, losses, every
    parameter gradient, and - 'bn' - the running_mean / running_var / num_batches_tracked of all 47 BatchNorm1d layers of the point encoder
    after the step (batch statistics, momentum 0.1, unbiased variance).  'bn': batch statistics make the synthetic-weight network's gradients
    ill-conditioned in fp32 - the reference's own fp32 gradients sit a median 1.8e-3 (up to 6e-2) from its fp64 ones (g_err32) - and the
    forward too (two fp32 evaluations differ by 1e-4 at the last encoder stage: tools/diag_norm_train.py shows the HIP path and the
    oracle's fp32 evaluation equally far from fp64, block by block).  That forward difference flips ReLUs whose input is ~0 (the point
    score head holds pre-ReLU values of 3e-5 ... 3e-4 in rows that carry a loss gradient: tools/diag_norm_mid.py traced the whole
    downstream deviation to ONE such row), which changes that row's gradient as a whole: parameters behind the encoder, where the
    reference's own fp32 run happens to agree with fp64 to 1e-5, see 1e-3 ... 4e-3 from any evaluation with another summation order.  So
    under 'bn' a parameter passes within 5x of max(its own g_err32, the median g_err32 of the configuration); 'ln' is judged like 'gn'."""
    from cofii2p_amd.network import CoFiI2P

    gold = load_golden("train_ref_%s.npz" % norm)
    assert str(gold["norm"]) == norm

    class OptN(Opt):
        pass

    OptN.norm = norm
    m = CoFiI2P(OptN(), arithmetic="bf16x6").to(DEV)
    m.train()
    worst, nbuf = _check_step_against(gold, m, GRAD_TOL, err32_mult=5.0 if norm == "bn" else 2.5, err32_floor="median" if norm == "bn" else 0.0)
    print("worst parameter (error / allowance) %.3g (%s); %d buffers compared" % (worst + (nbuf,)))
    assert nbuf == (36 + 3 * 47 if norm == "bn" else 36)   # 12 BatchNorm2d of the up-samplers (+ 47 BatchNorm1d), three buffers each
    # eval(): the same module on its (moved) running statistics still differentiates (BatchNorm as constants)
    m.eval()
    m.zero_grad()
    dd, img, batch, _ = _train_inputs(gold)
    outs = m(dd, img, batch["fine_center_kpt_coors"], batch["fine_xy"], batch["fine_pc_inline_index"], "train")
    sum(o.square().sum() for o in outs[:6]).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_train_step_bn_second_label_seed_statistical():
    """'bn' again, with the label seed the main fixture had to avoid (seed 5: a gradient-carrying row of the point score head holds a
    pre-ReLU value of 2.7e-5, which ANY two fp32 evaluations of the ill-conditioned batch-statistics forward disagree about - see
    test_train_step_other_point_norms_match_reference).  ADVICE r4: the seed must not be doing the work.  Forward outputs, losses and
    BatchNorm buffers are held to the same hard bounds as with seed 10; the gradients are judged statistically: the bulk of the parameters
    must sit inside the same allowance (>= 98 %, median error / allowance <= 0.5) and nothing may be far off (<= 3 x) - a flipped
    ReLU moves the rows behind it, it does not excuse a wrong backward.  Measured on MI355X: 99.6 % inside, median 0.115, worst 1.22 x
    (pc_score_layer.0.weight, the layer behind that row)."""
    from cofii2p_amd.network import CoFiI2P

    gold = load_golden("train_ref_bn_s5.npz")
    assert str(gold["norm"]) == "bn"

    class OptN(Opt):
        pass

    OptN.norm = "bn"
    m = CoFiI2P(OptN(), arithmetic="bf16x6").to(DEV)
    m.train()
    ratios = []
    _check_step_against(gold, m, GRAD_TOL, err32_mult=5.0, err32_floor="median", ratios=ratios)
    r = np.array(sorted(x for x, _ in ratios))
    inside = float((r < 1.0).mean())
    print("label seed 5: %d parameters, inside the allowance %.1f %%, median %.3g, p95 %.3g, worst %.3g (%s)" % (
        len(r), 100 * inside, float(np.median(r)), float(np.quantile(r, 0.95)), r[-1], max(ratios)[1]))
    assert inside >= 0.98 and float(np.median(r)) <= 0.5 and r[-1] <= 3.0


def _train_inputs(gold):
    fr, data = frame_inputs(int(gold["frame_id"]), int(gold["num_points"]), int(gold["pyr_seed"]))
    assert sha(fr.points) == str(gold["sha_points"]) and sha(fr.img) == str(gold["sha_img"])
    dd = {k: [t.to(DEV) for t in v] for k, v in data.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
    dd["feats"] = data["feats"].to(DEV)
    img = torch.from_numpy(fr.img)[None].to(DEV)
    batch = {k[4:]: G(gold[k]) for k in gold.files if k.startswith("lab_")}

    class StepOpt:
        dist_thres, pos_margin, neg_margin = float(gold["dist_thres"]), float(gold["pos_margin"]), float(gold["neg_margin"])

    return dd, img, batch, StepOpt


# "bf16x3" (opt-in for training: CoFiI2P(opt, arithmetic="bf16x3")): the split's 2^-16 per product is amplified by the cancellations of the
# backward to ~1e-2 in individual gradients - measured, documented in INTEGRATION.md, bounded here
@pytest.mark.parametrize("arith,tol", [("f32", GRAD_TOL), ("bf16x6", GRAD_TOL), (None, GRAD_TOL), ("bf16x3", 3e-2)])
def test_train_step_matches_reference(gold, arith, tol, monkeypatch):
    """forward(mode='train') -> losses -> backward: train-mode outputs, losses, every parameter gradient (norm-relative error of the 96
    recorded entries, gradient norm, sum), presence / absence of a gradient, BatchNorm running statistics - against the reference."""
    from cofii2p_amd import ops
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.train_step import step_losses

    monkeypatch.setattr(ops, "GEMM_MODE", arith or "bf16x3")   # arithmetic=None: the process default is bf16x3, training still computes fp32-grade (bf16x6, network.py forward())
    dd, img, batch, sopt = _train_inputs(gold)
    m = CoFiI2P(Opt(), arithmetic=arith).to(DEV)
    m.train()
    outs, mask, (l_desc, l_coarse, l_fine) = step_losses(m, dd, img, batch, sopt)
    assert np.array_equal(mask.cpu().numpy(), gold["mask"])
    for n_, t in zip(("img_desc", "pc_desc", "img_score", "pc_score", "patches", "fine_pc"), outs[:6]):
        ref = gold["train_" + n_]
        assert tuple(t.shape) == ref.shape, n_
        assert float((t.detach().cpu() - torch.from_numpy(ref)).abs().max()) < 1e-3, n_    # the forward's own budget (north star)
    for name, val in (("loss_desc", l_desc), ("loss_coarse", l_coarse), ("loss_fine", l_fine)):
        assert abs(float(val.detach()) - float(gold[name])) < 1e-3 * max(1.0, abs(float(gold[name]))), name
    (l_desc + l_coarse + l_fine).backward()
    names = [str(n) for n in gold["g_names"]]
    params = dict(m.named_parameters())
    assert names == list(params)
    # Judged against the reference's formulas evaluated in float64 (g_val64 / g_norm64).  The reference's own fp32 gradients are not a
    # usable yardstick everywhere: g_err32 records how far THEY are from the fp64 values - 2e-3 ... 5e-3 for the ResNet filters behind
    # InstanceNorm (ill-conditioned: a large common component cancels), ~1e-4 elsewhere.  A parameter passes when its gradient is within
    # `tol` of the fp64 gradient, or - where fp32 cannot do that - no further from it than the reference's own fp32 result was.
    total = float(np.sqrt((gold["g_norm64"] ** 2).sum()))
    zero_floor = 1e-6 * total   # biases in front of a one-channel-per-group GroupNorm: exactly zero gradient, fp32 leaves rounding noise
    worst = (0.0, None)
    beyond = 0
    for i, name in enumerate(names):
        p = params[name]
        if not int(gold["g_has"][i]):
            assert p.grad is None, "%s: the reference leaves this parameter without a gradient" % name
            continue
        assert p.grad is not None, name
        flat = p.grad.detach().double().reshape(-1).cpu()
        norm_ref = float(gold["g_norm64"][i])
        if norm_ref < zero_floor:
            # fp32 leaves rounding noise where the gradient is mathematically zero; the reference's own noise is g_norm (1.5e-3 under 'bn')
            assert float(flat.norm()) < max(1e-5 * total, 3.0 * float(gold["g_norm"][i])), "%s: mathematically zero gradient, got |g| = %.3g" % (name, float(flat.norm()))
            continue
        got = flat[torch.from_numpy(gold["g_pos"][i])].numpy()
        ref = gold["g_val64"][i]
        scale = max(np.linalg.norm(ref), norm_ref * math.sqrt(len(ref) / flat.numel()))   # the sampled entries' own size / the tensor's typical size
        e_samp = float(np.linalg.norm(got - ref) / scale)
        e_norm = abs(float(flat.norm()) - norm_ref) / norm_ref
        allow = max(tol, 2.5 * float(gold["g_err32"][i]))   # same rounding noise, another summation order: within 2.5x of the reference's own deviation
        beyond += e_samp > tol
        worst = max(worst, (max(e_samp, e_norm), name))
        assert e_samp < allow and e_norm < allow, "%s: sampled entries off by %.3g (allowed %.3g), norm by %.3g (relative)" % (name, e_samp, allow, e_norm)
    print("worst parameter gradient error %.3g (%s); %d parameters beyond %.0e, all of them where the reference's own fp32 gradient is" % (worst + (beyond, tol)))
    bufs = dict(m.named_buffers())
    for k in gold.files:
        if k.startswith("buf/"):
            b = bufs[k[4:]].detach().cpu()
            ref = torch.from_numpy(np.asarray(gold[k]))
            assert torch.allclose(b.double(), ref.double(), rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("train_bn,num_points,img_hw", [(False, 2048, (160, 512)), (True, 2048, (160, 512)), (True, 4096, (128, 320))])
def test_backward_matches_the_cpu_oracle_autograd(train_bn, num_points, img_hw):
    """Independent of the reference fixture: the differentiable forward - eval() mode (BatchNorm on running statistics) and train() mode
    (batch statistics) - against torch.autograd through the CPU oracle (oracle/cofi_oracle.py, the validated restatement of the reference forward),
    a linear probe of all six outputs as the loss, EVERY parameter's gradient.  Both sides are fp32 with different summation orders and the
    probe (a ramp over unit-norm descriptors) cancels heavily, so they agree to the fp32 noise level only: observed <= 6.1e-3 outside the
    image branch (DESIGN.md section 3a explains the conditioning; the reference fixture above is judged in float64 instead)."""
    import cofi_oracle as O
    import knn_c
    from cofii2p_amd import train_forward
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.spec import synth_state_dict
    from cofii2p_amd.synth import make_frame

    fr = make_frame(2, num_points=num_points, img_hw=img_hw)
    pyr = O.build_pyramid(np.ascontiguousarray(fr.points.T), 5, np.random.RandomState(4), knn=knn_c.knn_torch_compatible)
    data = dict(pyr)
    data["feats"] = torch.from_numpy(fr.feats)
    img = torch.from_numpy(fr.img)[None]
    g = np.random.default_rng(0)
    kpt = torch.from_numpy(np.stack([g.integers(2, img_hw[1] // 2 - 2, 8), g.integers(2, img_hw[0] // 2 - 2, 8)]).astype(np.float32))
    inl = torch.from_numpy(g.integers(0, num_points // 2, 8))

    class OptHW(Opt):
        img_H, img_W = img_hw

    probe = lambda outs: sum((o * torch.linspace(-1, 1, o.numel(), device=o.device).reshape(o.shape)).sum() for o in outs[:6])
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict().items()}
    leaves = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "kernel_points"))}
    sd = {k: v.clone() for k, v in sd.items()}   # train_bn updates the running buffers in place
    sd.update(leaves)
    probe(O.forward(sd, data, img, kpt, inl, "val", train_bn=train_bn)).backward()
    m = CoFiI2P(OptHW(), arithmetic="f32").to(DEV)
    m.train(train_bn)
    dd = {k: [t.to(DEV) for t in v] for k, v in data.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
    dd["feats"] = data["feats"].to(DEV)
    probe(train_forward.forward_train(m, dd, img.to(DEV), kpt.to(DEV), inl.to(DEV))).backward()
    total = math.sqrt(sum(float(v.grad.double().pow(2).sum()) for v in leaves.values() if v.grad is not None))
    worst, loose, errs = (0.0, None), 0, []
    for name, p in m.named_parameters():
        ref = leaves[name].grad
        if ref is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        if float(ref.norm()) < 1e-6 * total:   # exactly-zero gradients (bias in front of a per-channel normalisation): rounding noise on both sides
            assert float(p.grad.norm()) < 1e-5 * total, name
            continue
        e = rel_err(p.grad, ref)
        worst = max(worst, (e, name))
        errs.append((e, name))
        loose += e > GRAD_TOL
    errs.sort(reverse=True)
    print("worst gradient deviation from the oracle's autograd %.3g (%s); %d of %d parameters beyond 1e-3; top: %s"
          % (worst + (loose, len(errs), ", ".join("%s %.1e" % (n.replace("pc_encoder.", "pc.").replace("img_encoder.backbone.", "rn."), e) for e, n in errs[:12]))))
    for e, name in errs:
        ill = name.startswith("img_encoder.") or name.startswith("img_upsample")   # behind InstanceNorm / BatchNorm over a whole map
        assert e < (2e-2 if ill else 1e-2), (name, e)   # a wrong adjoint shows up as O(1); what is left here is fp32 summation-order noise


def test_training_changes_the_served_weights(gold):
    """train.py's loop: optimisation steps, then the validation pass (model.eval(), mode='val' under no_grad, train.py:66-70) must see
    the UPDATED weights - the packed / folded inference weights are re-derived once the parameters' versions moved - and the loss of
    the same batch goes down over a few Adam steps."""
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.train_step import step_losses, train_step

    dd, img, batch, sopt = _train_inputs(gold)
    m = CoFiI2P(Opt(), arithmetic="f32").to(DEV)
    kpt, inl = batch["fine_center_kpt_coors"], batch["fine_pc_inline_index"]
    m.eval()
    with torch.no_grad():
        before = [t.clone() for t in m(dd, img, kpt, None, inl, "val")[:6]]
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-3)   # train.py:163-164
    losses = [float(sum(train_step(m, opt, dd, img, batch, sopt))) for _ in range(4)]
    assert losses[-1] < losses[0], losses
    m.eval()
    with torch.no_grad():
        after = m(dd, img, kpt, None, inl, "val")[:6]
    assert any(float((a - b).abs().max()) > 1e-4 for a, b in zip(before, after))
    # the differentiable forward in eval() mode (BatchNorm on its running statistics) is the same function as the inference path
    from cofii2p_amd import train_forward

    graph = train_forward.forward_train(m, dd, img, kpt, inl)[:6]
    assert graph[0].requires_grad
    for a, b, n_ in zip(after, graph, ("img_desc", "pc_desc", "img_score", "pc_score", "patches", "fine_pc")):
        assert float((a - b.detach()).abs().max()) < 2e-4, n_


@pytest.mark.parametrize("fused", [False, True])
def test_graphed_train_step_equals_eager(gold, fused):
    """train_step.GraphedTrainStep (forward + losses + backward + Adam as ONE hipGraph, replayed per frame) against the eager step with
    the same capturable optimizer: two alternating frames of one signature (the static inputs are restaged every call), a learning-rate
    change the way train.py:326-330 makes it, then the validation forward on the updated weights.  The step's kernels (incl. the
    package's own upsample2x backward) are bit-reproducible launch by launch, but the recording runs its branches on other streams than the
    eager step and Adam turns a last-bit difference of a near-zero gradient into a visible one - hence "losses to 1e-4, all but a sliver of the parameters to 1e-5" rather than bit equality."""
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.train_step import GraphedTrainStep, train_step

    dd, img, batch, sopt = _train_inputs(gold)
    fr2, data2 = frame_inputs(int(gold["frame_id"]) + 1, int(gold["num_points"]), int(gold["pyr_seed"]) + 1)
    dd2 = {k: [t.to(DEV) for t in v] for k, v in data2.items() if k in ("points", "neighbors", "subsampling", "upsampling")}
    dd2["feats"] = data2["feats"].to(DEV)
    img2 = torch.from_numpy(fr2.img)[None].to(DEV)
    batch2 = {k: (v.roll(5, -1) if v.dim() and v.shape[-1] == batch["pc_kpt_idx"].numel() and k not in ("K_4", "P") else v) for k, v in batch.items()}
    frames = [(dd, img, batch), (dd2, img2, batch2)]
    sd0 = {k: v.clone() for k, v in CoFiI2P(Opt(), arithmetic="bf16x6").state_dict().items()}

    def run(graphed):
        m = CoFiI2P(Opt(), arithmetic="bf16x6").to(DEV)
        m.load_state_dict(sd0)
        # (the learning rate as a device scalar on both sides: Adam divides by float32(lr) then, not by the double - one ulp apart)
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=torch.full((), 1e-3, device=DEV), capturable=True, fused=fused)
        step = GraphedTrainStep(m, opt, sopt) if graphed else None
        losses = []
        for it in range(7):
            pc, im, b = frames[it % 2]
            if it == 5:
                for g in opt.param_groups:
                    g["lr"] = 2.5e-4 if graphed else torch.full((), 2.5e-4, device=DEV)   # train.py:329-330 (a float: the recording's scalar follows)
            ls = step(pc, im, b) if graphed else torch.stack(train_step(m, opt, pc, im, b, sopt))
            losses.append(ls.cpu().numpy().copy())
        if graphed:
            assert step.replays == 6 and m._replayed_steps == 6              # call 1 eager, call 2 records + replays
        m.eval()
        with torch.no_grad():
            val = [t.clone() for t in m(dd, img, batch["fine_center_kpt_coors"], None, batch["fine_pc_inline_index"], "val")[:6]]
        return np.stack(losses), {k: v.detach().clone() for k, v in m.state_dict().items()}, val

    l_e, p_e, v_e = run(False)
    l_2, p_2, _v2 = run(False)
    l_g, p_g, v_g = run(True)
    assert np.abs(l_g - l_e).max() <= 1e-4 * np.abs(l_e).max(), (l_e, l_g)
    assert (np.abs(np.diff(l_e.sum(1))) > 1e-3).all()                            # the frames alternate and the weights move: no two steps alike
    moved = sum(float((p_e[k].float() - sd0[k].to(DEV).float()).abs().max()) > 1e-4 for k in p_e)
    assert moved > 100                                                          # most of the 430 tensors took Adam steps

    def apart(pa, pb):
        return sum(int(((pa[k].double() - pb[k].double()).abs() > 1e-5).sum()) for k in pa)

    n_all = sum(v.numel() for v in p_e.values())
    noise = apart(p_e, p_2)                                                     # two eager runs of the same seven steps: torch's atomics
    n_off = apart(p_e, p_g)
    print("elements further apart than 1e-5 after 7 steps: eager/eager %d, eager/graphed %d of %d" % (noise, n_off, n_all))
    assert n_off <= max(3 * noise, 1e-3 * n_all), (n_off, noise, n_all)
    for a, b in zip(v_e, v_g):
        assert float((a - b).abs().max()) < 1e-3
    with pytest.raises(ValueError):
        GraphedTrainStep(CoFiI2P(Opt()).to(DEV), torch.optim.Adam([torch.nn.Parameter(torch.zeros(1, device=DEV))], lr=1e-3), sopt)


def test_train_mode_refusals():
    from cofii2p_amd import _lib
    from cofii2p_amd.network import CoFiI2P

    class OptBn(Opt):
        norm = "bn"

    dd = {"points": [torch.zeros((8, 3), device=DEV)], "neighbors": [], "subsampling": [], "upsampling": [], "feats": torch.zeros((8, 4), device=DEV)}
    m = CoFiI2P(OptBn()).to(DEV)
    m.train()
    with pytest.raises(NotImplementedError), torch.no_grad():   # the folded inference sequence has no batch statistics: eval(), or mode 'train' / 'val'
        m(dd, torch.zeros((1, 3, 160, 512), device=DEV), None, None, None, "test")
    m = CoFiI2P(Opt()).to(DEV)
    with pytest.raises(ValueError):
        m(dd, torch.zeros((2, 3, 160, 512), device=DEV), None, None, None, "train")
    with pytest.raises(NotImplementedError):   # mode='test' has no gradient to give
        m(dd, torch.zeros((1, 3, 160, 512), device=DEV, requires_grad=True), None, None, None, "test")
    with pytest.raises(_lib.CofiError):
        m(dd, torch.zeros((1, 3, 160, 512)), None, None, None, "train")
