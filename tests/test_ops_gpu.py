"""Parity of every HIP kernel (through the C ABI) against the CPU oracle / golden vectors.
Needs a real MI355X:  python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cofi_oracle as O  # noqa: E402
import knn_c  # noqa: E402
from common import load_golden  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    """The kernel tests of this file compare against fp64 / reference values at fp32 tolerances: they run the exact-fp32 GEMM
    arithmetic unless a test selects "bf16x3" (the library default) itself."""
    from cofii2p_amd import ops as _ops

    assert torch.cuda.is_available()
    saved, _ops.GEMM_MODE = _ops.GEMM_MODE, "f32"
    yield _ops
    _ops.GEMM_MODE = saved


@pytest.fixture(scope="module")
def mg():
    return load_golden("micro_ops.npz")


def G(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def close(a, b, tol=1e-4):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (128, 128, 128), (1280, 512, 7680), (77, 33, 60), (1280, 1, 64), (2560, 1024, 3072),
                                   (20480, 32, 64), (300, 200, 36), (5, 7, 4),
                                   # XCD-contiguous tile order: 2 / 4 / 8 / 16 column tiles x a row-tile count that is not a multiple of 8
                                   (2400, 256, 128), (4700, 512, 64), (1350, 1024, 256), (2368 + 13, 2048, 64)])
def test_gemm(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    rd = torch.randint(1, 9, (M,), generator=g).float()
    ref = (a.double() @ w.double().t())
    close(ops.gemm(G(a), G(w)), ref.float(), 2e-4)
    close(ops.gemm(G(a), G(w), bias=G(bias), rowdiv=G(rd), act=ops.ACT_RELU), torch.relu(ref / rd[:, None].double() + bias.double()).float(), 2e-4)
    close(ops.gemm(G(a), G(w), bias=G(bias), act=ops.ACT_SIGMOID), torch.sigmoid(ref + bias.double()).float(), 2e-4)


@pytest.mark.parametrize("M,N,K,groups", [(1280, 128, 128, 128), (1000, 64, 480, 32), (1280, 2048, 512, 32), (20480, 32, 128, 32), (77, 96, 36, 96),
                                          (4700, 512, 64, 32), (1350, 1024, 256, 32)])  # re-numbered tiles: slab id = remapped row tile
def test_gemm_fused_column_statistics(ops, M, N, K, groups):
    g = torch.Generator().manual_seed(M + N)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    rd = torch.randint(1, 5, (M,), generator=g).float()
    y, part = ops.gemm_colstats(G(a), G(w), bias=G(b), rowdiv=G(rd))
    ref = ((a.double() @ w.double().t()) / rd[:, None].double() + b.double())
    close(y, ref.float(), 2e-4)
    st = ops.group_stats_from_colpart(part, M, groups).cpu()
    gr = ref.reshape(M, groups, N // groups)
    mean, var = gr.mean((0, 2)), gr.var((0, 2), unbiased=False)
    close(st[:, 0], mean.float(), 1e-4)
    close(st[:, 1], torch.rsqrt(var + 1e-5).float(), 1e-4)
    close(ops.col_inv_norm_from_colpart(part, M, min(N, 32)), (1.0 / ref[:, : min(N, 32)].norm(dim=0)).float(), 1e-4)


@pytest.mark.parametrize("M,N,K", [(1280, 128, 128), (1280, 128, 256), (100, 64, 64), (33, 100, 36)])
def test_gemm_fused_layernorm(ops, M, N, K):
    g = torch.Generator().manual_seed(M + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    ga, be, r = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = torch.nn.functional.layer_norm(a @ w.t(), (N,), ga, be)
    close(ops.gemm_layernorm(G(a), G(w), G(ga), G(be)), ref, 5e-5)
    close(ops.gemm_layernorm(G(a), G(w), G(ga), G(be), relu=True, res=G(r)), torch.relu(ref) + r, 5e-5)
    buf = torch.zeros(M, 2 * N + 8, device=DEV)  # strided output view
    ops.gemm_layernorm(G(a), G(w), G(ga), G(be), out=buf[:, N + 8:])
    close(buf[:, N + 8:], ref, 5e-5)
    assert float(buf[:, : N + 8].abs().max()) == 0.0


def test_gemm_strided_views(ops):
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(200, 256, generator=g)
    w = torch.randn(96, 128, generator=g) * 0.1
    a = G(buf)
    out = torch.zeros(200, 160, device=DEV)
    ops.gemm(a[:, 128:], G(w), out=out[:, 32:128])
    close(out[:, 32:128], buf[:, 128:] @ w.t(), 2e-4)
    assert float(out[:, :32].abs().max()) == 0.0 and float(out[:, 128:].abs().max()) == 0.0


def test_kpconv_micro_golden(ops, mg):
    idx = G(mg["kp_idx"], torch.int32)
    agg, cnt = ops.kpconv_aggregate(G(mg["kp_feats"]), G(mg["kp_q_pts"]), G(mg["kp_s_pts"]), idx, G(mg["kp_kernel_points"]), 0.2)
    w = torch.from_numpy(mg["kp_weights"])  # (15,Cin,Cout) -> (Cout, 15*Cin)
    wp = w.permute(2, 0, 1).reshape(w.shape[2], -1).contiguous()
    out = ops.gemm(agg, G(wp), bias=G(mg["kp_bias"]), rowdiv=cnt)
    close(out, mg["kp_out"], 2e-5)


@pytest.mark.parametrize("N,M,H,C,Co", [(500, 300, 128, 32, 32), (2000, 1000, 128, 4, 64), (700, 350, 128, 128, 128), (400, 200, 64, 512, 64),
                                        (300, 200, 64, 3, 16), (300, 130, 128, 1, 8), (900, 500, 256, 4, 32)])
def test_kpconv_random(ops, N, M, H, C, Co):
    g = np.random.default_rng(N + C)
    s_pts = torch.from_numpy(g.uniform(-1, 1, (N, 3)).astype(np.float32))
    q_pts = s_pts[g.integers(0, N, M)] + 0.01
    idx = O.knn_torch(s_pts, q_pts, H)
    idx[::11, -5:] = N
    feats = torch.from_numpy(g.standard_normal((N, C)).astype(np.float32))
    feats[::9] = 0
    kp = torch.from_numpy((g.uniform(-0.3, 0.3, (15, 3))).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((15, C, Co)) * 0.1).astype(np.float32))
    b = torch.from_numpy(g.standard_normal(Co).astype(np.float32))
    ref = O.kpconv(feats, q_pts, s_pts, idx, kp, w, b, 0.35)
    agg, cnt = ops.kpconv_aggregate(G(feats), G(q_pts), G(s_pts), G(idx, torch.int32), G(kp), 0.35)
    wp = w.permute(2, 0, 1).reshape(Co, -1).contiguous()
    if (15 * C) % 4 == 0:
        out = ops.gemm(agg, G(wp), bias=G(b), rowdiv=cnt)
    else:   # the GEMM kernels want K % 4 == 0: finish odd channel counts on the host
        out = (agg.cpu().double() @ wp.double().T / cnt.cpu().double()[:, None] + b.double()).float()
    close(out, ref, 1e-4)
    if C % 4 == 0 and C > 4:
        # the aggregate written as bf16 hi / lo planes (same rounding as the GEMM's own on-the-fly split) + the LDS-DMA GEMM that takes
        # both operands as planes (gemm_planes_kernel): at equal tile rows / split-K the result equals the fp32-aggregate path through
        # the register-staged kernel bit for bit; with the default plans of each it stays within the oracle tolerance
        import ctypes

        lib = ops._lib.load()
        fp, fo = lib.cofi_tune_force_planes, lib.cofi_tune_force_plan   # tuning hooks (not part of the public header)
        fp.argtypes, fp.restype, fo.argtypes, fo.restype = [ctypes.c_int] * 2, ctypes.c_int, [ctypes.c_int] * 3, ctypes.c_int
        saved, ops.GEMM_MODE = ops.GEMM_MODE, "bf16x3"
        try:
            wsp = ops.presplit(G(wp))
            pl, cnt_p = ops.kpconv_aggregate(G(feats), G(q_pts), G(s_pts), G(idx, torch.int32), G(kp), 0.35, planes=True)
            assert isinstance(pl, ops.SplitA) and torch.equal(cnt_p, cnt)
            hi = (pl.planes[0].to(torch.int32) << 16).view(torch.float32)
            lo = (pl.planes[1].to(torch.int32) << 16).view(torch.float32)
            assert float((hi + lo - agg).abs().max()) <= 2e-5 * max(1.0, float(agg.abs().max()))
            for ks in (1, 3):
                if ks > 1 and 15 * C < 1024:
                    continue
                fo(64, 64, ks), fp(1, ks)   # 64 x 64 tiles, 4 waves, the same K chunks in both kernels
                y1, part1 = ops.gemm_colstats(agg, wsp, bias=G(b), rowdiv=cnt, stat_width=1)
                y2, part2 = ops.gemm_colstats(pl, wsp, bias=G(b), rowdiv=cnt_p, stat_width=1)
                assert torch.equal(y2, y1) and torch.equal(part2, part1), ks
            fo(0, 0, 0), fp(-1, 0)
            close(ops.gemm(pl, wsp, bias=G(b), rowdiv=cnt_p), ref, 1e-4)
        finally:
            fo(0, 0, 0), fp(-1, 0)
            ops.GEMM_MODE = saved
    if C <= 4:   # the first-layer form (32-byte records) against the general kernel fed with the same flags: identical bits
        agg_g, cnt_g = ops.kpconv_aggregate(G(feats), G(q_pts), G(s_pts), G(idx, torch.int32), G(kp), 0.35, row_pos=ops.row_sum_positive(G(feats)))
        assert torch.equal(agg_g, agg) and torch.equal(cnt_g, cnt)


@pytest.mark.parametrize("N,M,H,C,frames,use_order", [(900, 320, 128, 64, 1, False), (1200, 512, 128, 32, 1, True), (5000, 16384, 128, 32, 1, True),
                                                      (4000, 8192, 64, 64, 1, True), (600, 256, 128, 64, 2, True), (300, 16, 128, 32, 1, False)])
def test_kpconv_fused_vs_oracle_and_two_kernel_path(ops, monkeypatch, N, M, H, C, frames, use_order):
    """cofi_kpconv_fused (aggregate + bf16x3 GEMM + statistics partials in one kernel, kpconv.py:91-116) against the oracle, against
    the aggregate + GEMM path it replaces, its partials against sums of its own output, and its partials as consumed by a
    normalising GEMM loader (slabs of 64 / 32 / 16 rows)."""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x3")
    g = np.random.default_rng(N + C + M)
    ys, refs = [], []
    s_all, q_all, i_all, f_all = [], [], [], []
    kp = torch.from_numpy((g.uniform(-0.3, 0.3, (15, 3))).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((15, C, C)) * 0.1).astype(np.float32))
    b = torch.from_numpy(g.standard_normal(C).astype(np.float32))
    for f in range(frames):
        s_pts = torch.from_numpy(g.uniform(-1, 1, (N, 3)).astype(np.float32))
        q_pts = s_pts[g.integers(0, N, M)] + 0.01
        idx = torch.from_numpy(knn_c.knn(s_pts.numpy(), q_pts.numpy(), H).astype(np.int64))
        idx[::11, -5:] = N          # shadow neighbours
        idx[5, :] = N               # a query without any neighbour: bias only
        feats = torch.from_numpy(g.standard_normal((N, C)).astype(np.float32))
        feats[::9] = 0
        refs.append(O.kpconv(feats, q_pts, s_pts, idx, kp, w, b, 0.35))
        s_all.append(s_pts), q_all.append(q_pts), i_all.append(idx), f_all.append(feats)
    S, Q, I, F = (torch.cat(t) for t in (s_all, q_all, i_all, f_all))
    ref = torch.cat(refs)
    wp = ops.presplit(G(w.permute(2, 0, 1).reshape(C, -1).contiguous()))
    order = None
    if use_order:   # any permutation of the queries of a frame is a valid processing order
        order = G(np.concatenate([g.permutation(M) for _ in range(frames)]).astype(np.int32))
    width = 2 if C == 64 else 1
    y, part, sr = ops.kpconv_fused(G(F), G(Q), G(S), G(I, torch.int32), G(kp), 0.35, wp, G(b), stat_width=width, frames=frames, order=order)
    assert sr == ops.kpconv_fused_slab_rows(C, M, frames) and sr in (16, 32, 64) and part.shape == (M * frames // sr, C // width, 2)
    close(y, ref, 1e-4)
    assert torch.allclose(y[5].cpu(), b)
    agg, cnt = ops.kpconv_aggregate(G(F), G(Q), G(S), G(I, torch.int32), G(kp), 0.35, frames=frames)
    two = ops.gemm(agg, wp, bias=G(b), rowdiv=cnt)
    assert float((y - two).abs().max()) < 2e-5 * max(1.0, float(two.abs().max()))
    # partials: per frame, the sum over its slabs = column (group) sums of y
    yd = y.double().reshape(frames, M, C // width, width)
    pd = part.double().reshape(frames, -1, C // width, 2).sum(1)
    assert torch.allclose(pd[..., 0], yd.sum((1, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(pd[..., 1], (yd * yd).sum((1, 3)), rtol=1e-5, atol=1e-3)
    # ... and a normalising GEMM loader folds them like any other table
    if M % 128 == 0 or frames == 1:
        st = ops.ColStats(part, M * frames, 32, frames, width=width, slab_rows=sr)
        gam, bet = G(g.standard_normal(C).astype(np.float32)), G(g.standard_normal(C).astype(np.float32))
        w2 = ops.presplit(G((g.standard_normal((48, C)) * 0.2).astype(np.float32)))
        fused = ops.gemm(ops.Normed(y, st, gam, bet, 0.1), w2, frames=frames)
        yn = torch.cat([O.group_norm_rows(y[f * M:(f + 1) * M].cpu(), gam.cpu(), bet.cpu()) for f in range(frames)])
        want = torch.nn.functional.leaky_relu(yn, 0.1) @ w2.w.cpu().T
        close(fused, want, 2e-4 if M >= 256 else 1e-3)   # 16 rows: the statistics amplify the bf16x3 rounding of y


def test_kpconv_fused_rejects_unsupported_shapes(ops, monkeypatch):
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x3")
    assert ops.kpconv_fused_slab_rows(128, 1024) == 0 and ops.kpconv_fused_slab_rows(64, 1000) == 0
    assert ops.kpconv_fused_slab_rows(64, 20480) == 64 and ops.kpconv_fused_slab_rows(64, 10240) == 32 and ops.kpconv_fused_slab_rows(32, 5120) == 16
    assert ops.kpconv_fused_slab_rows(64, 20480, 16) == 64 and ops.kpconv_fused_slab_rows(64, 2560, 16) == 64
    monkeypatch.setattr(ops, "GEMM_MODE", "f32")
    assert ops.kpconv_fused_slab_rows(64, 20480) == 0


@pytest.mark.parametrize("sigma", [0.01, 0.08, 3.0])
@pytest.mark.parametrize("C,H", [(4, 128), (32, 128), (64, 256), (128, 64)])
def test_kpconv_support_cut_extremes(ops, C, H, sigma):
    """The aggregation drops neighbours outside the kernel's support (|d| >= max|k| + sigma: every influence exactly 0).  Extremes:
    a support so small that only coincident points are inside (incl. queries with NO neighbour inside), one that holds every
    neighbour, several staging phases (H = 256); kernel points on and beyond the unit scale."""
    g = np.random.default_rng(C + H)
    N, M = 1500, 400
    s_pts = torch.from_numpy(g.uniform(-1, 1, (N, 3)).astype(np.float32))
    q_pts = s_pts[g.integers(0, N, M)].clone()
    q_pts[::2] += 0.05          # half of the queries coincide with a support point, the others sit 0.087 away from everything near
    idx = torch.from_numpy(knn_c.knn(s_pts.numpy(), q_pts.numpy(), H).astype(np.int64))
    idx[::7, -9:] = N
    feats = torch.from_numpy(g.standard_normal((N, C)).astype(np.float32))
    kp = torch.from_numpy((g.uniform(-1, 1, (15, 3)) * 1.5 * sigma).astype(np.float32))
    kp[0] = 0
    w = torch.from_numpy((g.standard_normal((15, C, 16)) * 0.1).astype(np.float32))
    b = torch.from_numpy(g.standard_normal(16).astype(np.float32))
    ref = O.kpconv(feats, q_pts, s_pts, idx, kp, w, b, sigma)
    agg, cnt = ops.kpconv_aggregate(G(feats), G(q_pts), G(s_pts), G(idx, torch.int32), G(kp), sigma)
    out = ops.gemm(agg, G(w.permute(2, 0, 1).reshape(16, -1).contiguous()), bias=G(b), rowdiv=cnt)
    close(out, ref, 1e-4)
    if sigma == 0.01:
        # (almost) nothing lies inside the support of the shifted queries: exact zero rows, bias-only outputs
        assert float((agg[::2].abs().amax(1) == 0).float().mean()) > 0.8


def test_pool_gather(ops, mg):
    idx = G(mg["kp_idx"], torch.int32)
    x = G(mg["pool_x"])
    assert torch.equal(ops.neighbor_maxpool(x, idx).cpu(), torch.from_numpy(mg["pool_out"]))
    assert torch.equal(ops.gather_rows(x, idx).cpu(), torch.from_numpy(mg["up_out"]))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 200, generator=g)
    idx = torch.randint(0, 1001, (333, 128), generator=g)
    assert torch.equal(ops.neighbor_maxpool(G(x), G(idx, torch.int32)).cpu(), O.neighbor_maxpool(x, idx))
    assert torch.equal(ops.gather_rows(G(x), G(idx, torch.int32)).cpu(), O.nearest_upsample(x, idx))


@pytest.mark.parametrize("M,C,groups", [(77, 64, 32), (1000, 32, 32), (333, 2048, 32), (1280, 128, 128), (130, 256, 32)])
def test_group_norm(ops, M, C, groups):
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 2 + 0.7
    ga, be = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = O.group_norm_rows(x, ga, be, groups)
    close(ops.group_norm(G(x), groups, G(ga), G(be)), ref, 2e-5)
    close(ops.group_norm(G(x), groups, G(ga), G(be), slope=0.1), O.leaky(ref), 2e-5)
    r = torch.randn(M, C, generator=g)
    st = ops.group_stats(G(x), groups)
    close(ops.group_norm_apply(G(x), st, G(ga), G(be), slope=0.1, res=G(r)), O.leaky(ref + r), 2e-5)
    rst = ops.group_stats(G(r), groups)
    close(ops.group_norm_apply(G(x), st, G(ga), G(be), slope=0.1, res=G(r), res_stats=rst, res_gamma=G(be), res_beta=G(ga)),
          O.leaky(ref + O.group_norm_rows(r, be, ga, groups)), 2e-5)
    # gamma-less, group width 1: the InstanceNorm + ReLU of the score heads
    if groups == C:
        var, mean = torch.var_mean(x, dim=0, unbiased=False, keepdim=True)
        close(ops.group_norm(G(x), groups, slope=0.0), torch.relu((x - mean) * torch.rsqrt(var + 1e-5)), 2e-5)


@pytest.mark.parametrize("frames,Mf,C,groups", [(16, 80, 512, 512), (3, 77, 64, 32), (5, 200, 256, 32)])
def test_group_stats_stacked_frames(ops, frames, Mf, C, groups):
    """stack mode with rows-per-frame that are NOT a multiple of any statistics slab (the 5x16 map of ResNet layer4)"""
    g = torch.Generator().manual_seed(frames + Mf + C)
    x = torch.randn(frames * Mf, C, generator=g) * 2 + 0.3
    st = ops.group_stats(G(x), groups, frames=frames)
    out = ops.group_norm_apply(G(x), st, slope=0.0, frames=frames).cpu()
    for f in range(frames):
        xf = x[f * Mf:(f + 1) * Mf]
        ref = torch.relu(O.group_norm_rows(xf, torch.ones(C), torch.zeros(C), groups))
        close(out[f * Mf:(f + 1) * Mf], ref, 2e-5)


@pytest.mark.parametrize("frames,Mf,C,groups,K", [(1, 1280, 128, 32, 128), (1, 320, 256, 256, 64), (4, 640, 64, 32, 96), (2, 1280, 512, 32, 64),
                                                  (1, 20480, 64, 32, 64)])
def test_group_norm_apply_from_column_partials(ops, frames, Mf, C, groups, K):
    """statistics folded inside the apply kernel (small tables) or by the finalize kernel (large): both equal the oracle's
    GroupNorm of the GEMM output; row_pos = (row sum > 0) comes out of the same kernel"""
    g = torch.Generator().manual_seed(frames * 131 + Mf + C)
    a, w = torch.randn(frames * Mf, K, generator=g), torch.randn(C, K, generator=g) / K ** 0.5
    r = torch.randn(frames * Mf, C, generator=g)
    ga, be = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    y, part = ops.gemm_colstats(G(a), G(w))
    yr, rpart = ops.gemm_colstats(G(r), G(torch.eye(C)))          # the residual with its own statistics
    st, rst = ops.ColStats(part, frames * Mf, groups, frames), ops.ColStats(rpart, frames * Mf, groups, frames)
    out = ops.group_norm_apply(y, st, G(ga), G(be), slope=0.1, res=yr, res_stats=rst, res_gamma=G(be), res_beta=G(ga), frames=frames,
                               want_row_pos=True)
    yc, outc = y.cpu(), out.cpu()
    for f in range(frames):
        sl = slice(f * Mf, (f + 1) * Mf)
        ref = O.leaky(O.group_norm_rows(yc[sl], ga, be, groups) + O.group_norm_rows(r[sl], be, ga, groups))
        close(outc[sl], ref, 3e-5)
    if C <= 256:
        rp = out.cofi_row_pos.cpu().bool()
        s = outc.double().sum(1)
        firm = s.abs() > 1e-4                                        # rows whose sum is not within rounding of zero
        assert torch.equal(rp[firm], (s > 0)[firm])
        assert torch.equal(out.cofi_row_pos, ops.row_sum_positive(out)) or (rp != ops.row_sum_positive(out).cpu().bool()).sum() <= 2
    else:
        assert not hasattr(out, "cofi_row_pos")


def test_group_norm_golden(ops, mg):
    close(ops.group_norm(G(mg["gn_x"]), 32, G(mg["gn_w"]), G(mg["gn_b"])), mg["gn_out"], 2e-5)
    y = ops.gemm(G(mg["un_x"]), G(mg["un_w"]), bias=G(mg["un_b"]))
    close(ops.group_norm(y, 32, G(mg["un_gw"]), G(mg["un_gb"]), slope=0.1), mg["un_out"], 2e-5)


@pytest.mark.parametrize("M,C", [(50, 128), (1280, 1024), (7, 512), (300, 256)])
def test_layer_norm(ops, M, C):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=g) * 3 + 1
    ga, be, r = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(M, C, generator=g)
    ref = torch.nn.functional.layer_norm(x, (C,), ga, be)
    close(ops.layer_norm(G(x), G(ga), G(be)), ref, 2e-5)
    close(ops.layer_norm(G(x), G(ga), G(be), relu=True), torch.relu(ref), 2e-5)
    close(ops.layer_norm(G(x), G(ga), G(be), res=G(r)), ref + r, 2e-5)


@pytest.fixture(params=["bf16x6", "bf16x6-presplit", "f32"])
def attn(request, ops):
    """`ops` with the attention kernel pinned to one arithmetic: the fp32-grade bf16 split (csrc/attention_x6.inc; K / V split by every
    workgroup, or ONCE by cofi_attention_kv_planes - "presplit") / the exact fp32 matrix instruction (csrc/attention.hip) - all held to
    the same tolerances"""
    old, old_rows = ops.ATTN_MODE, ops.ATTN_PRESPLIT_ROWS
    ops.ATTN_MODE = request.param.split("-")[0]
    ops.ATTN_PRESPLIT_ROWS = 1 if request.param.endswith("presplit") else 0
    yield ops
    ops.ATTN_MODE, ops.ATTN_PRESPLIT_ROWS = old, old_rows


@pytest.mark.parametrize("frames,L,S", [(1, 1280, 1280), (3, 700, 333), (1, 64, 2000), (2, 100, 4100), (1, 33, 31), (16, 1280, 1280), (1, 96, 5)])
def test_attention_presplit_is_bit_equal(ops, monkeypatch, frames, L, S):
    """cofi_attention_kv_planes + cofi_attention_parts_planes == cofi_attention_parts_bf16x6 bit for bit (same planes, same products, same order):
    full and partial key blocks, key ranges that end inside a step, K / V as strided views of one projection output"""
    monkeypatch.setattr(ops, "ATTN_MODE", "bf16x6")
    g = torch.Generator().manual_seed(3 * S + L)
    q = G(torch.randn(frames * L, 128, generator=g))
    kv = G(torch.randn(frames * S, 256, generator=g) * 1.5)
    cs = G(torch.rand(frames, 128, generator=g) + 0.5)
    outs = []
    for rows in (0, 1):
        monkeypatch.setattr(ops, "ATTN_PRESPLIT_ROWS", rows)
        outs.append(ops.attention(q, kv[:, :128], kv[:, 128:], q_colscale=cs, frames=frames).clone())
    assert torch.equal(outs[0], outs[1])


def test_attention_golden(attn, mg):
    q, k, v = (torch.from_numpy(mg[n]).reshape(-1, 128) for n in ("att_q", "att_k", "att_v"))
    close(attn.attention(G(q), G(k), G(v)), mg["att_out"].reshape(-1, 128), 2e-5)


@pytest.mark.parametrize("L,S", [(1280, 1280), (1280, 128), (100, 1000), (33, 5), (2560, 640)])
def test_attention_random(attn, L, S):
    g = torch.Generator().manual_seed(L + S)
    q, k, v = torch.randn(L, 128, generator=g), torch.randn(S, 128, generator=g) * 2, torch.randn(S, 128, generator=g)
    cs = torch.rand(128, generator=g) + 0.5
    ref = O.full_attention((q * cs).view(L, 4, 32), k.view(S, 4, 32), v.view(S, 4, 32)).reshape(L, 128)
    close(attn.attention(G(q), G(k), G(v), q_colscale=G(cs)), ref, 5e-5)


@pytest.mark.parametrize("frames,L,S", [(4, 1280, 1280), (8, 1280, 1280), (16, 320, 200)])
def test_attention_stacked_frames(attn, frames, L, S):
    """stack mode: frame f attends only to its own keys; large grids run with 4 / 2 key splits per workgroup"""
    g = torch.Generator().manual_seed(frames * 7 + L)
    q, k, v = torch.randn(frames * L, 128, generator=g), torch.randn(frames * S, 128, generator=g) * 2, torch.randn(frames * S, 128, generator=g)
    cs = torch.rand(frames, 128, generator=g) + 0.5
    out = attn.attention(G(q), G(k), G(v), q_colscale=G(cs), frames=frames).cpu()
    for f in (0, frames // 2, frames - 1):
        ref = O.full_attention((q[f * L:(f + 1) * L] * cs[f]).view(L, 4, 32), k[f * S:(f + 1) * S].view(S, 4, 32),
                               v[f * S:(f + 1) * S].view(S, 4, 32)).reshape(L, 128)
        close(out[f * L:(f + 1) * L], ref, 5e-5)


@pytest.mark.parametrize("frames,L", [(1, 1280), (2, 1280), (8, 640)])
def test_attention_q_norm_from_column_partials(attn, frames, L):
    """the token-axis norm of Q folded inside the attention kernel from the projection's column partials == the explicit
    per-frame column scale"""
    g = torch.Generator().manual_seed(frames + L)
    x, w = torch.randn(frames * L, 128, generator=g), torch.randn(384, 128, generator=g) / 11.0
    qkv, part = attn.gemm_colstats(G(x), G(w))
    q, k, v = qkv[:, :128], qkv[:, 128:256], qkv[:, 256:]
    cs = attn.col_inv_norm_from_colpart(part, frames * L, 128, frames=frames)
    a = attn.attention(q, k, v, q_colscale=cs, frames=frames)
    b = attn.attention(q, k, v, q_colpart=part, frames=frames)
    close(b, a.cpu(), 1e-5)
    qc = qkv.cpu()
    f = frames - 1
    sl = slice(f * L, (f + 1) * L)
    qn = qc[sl, :128] / qc[sl, :128].norm(dim=0).clamp_min(1e-12)
    ref = O.full_attention(qn.reshape(L, 4, 32), qc[sl, 128:256].reshape(L, 4, 32), qc[sl, 256:].reshape(L, 4, 32)).reshape(L, 128)
    close(b[sl], ref, 5e-5)


def test_attention_forced_rescale(attn):
    """a key block arriving late with a much larger score forces the online-softmax rescale branch"""
    L, S = 64, 256
    g = torch.Generator().manual_seed(0)
    q, k, v = torch.randn(L, 128, generator=g), torch.randn(S, 128, generator=g) * 0.1, torch.randn(S, 128, generator=g)
    k[200] = q[5] * 4.0
    k[37] = q[9] * 6.0
    ref = O.full_attention(q.view(L, 4, 32), k.view(S, 4, 32), v.view(S, 4, 32)).reshape(L, 128)
    close(attn.attention(G(q), G(k), G(v)), ref, 5e-5)


@pytest.mark.parametrize("frames,L,S", [(3, 700, 333), (1, 64, 2000), (1, 2560, 130), (2, 100, 4100), (1, 33, 31), (16, 1280, 1280)])
def test_attention_bf16x6_against_the_fp32_instruction_kernel(ops, frames, L, S):
    """the split-arithmetic attention kernel over 1 ... 5 steps per key range, a partial last key block and 16 stacked frames: bit-reproducible
    from launch to launch, and within 5e-6 of the largest output of the exact-fp32-instruction kernel (both carry ~1e-6 of their own)"""
    g = torch.Generator().manual_seed(11 + S)
    q, k, v = G(torch.randn(frames * L, 128, generator=g)), G(torch.randn(frames * S, 128, generator=g) * 2), G(torch.randn(frames * S, 128, generator=g))
    old = ops.ATTN_MODE
    try:
        ops.ATTN_MODE = "bf16x6"
        a = ops.attention(q, k, v, frames=frames).clone()
        b = ops.attention(q, k, v, frames=frames).clone()
        ops.ATTN_MODE = "f32"
        ref = ops.attention(q, k, v, frames=frames)
    finally:
        ops.ATTN_MODE = old
    assert torch.equal(a, b)
    assert torch.isfinite(a).all()
    assert float((a - ref).abs().max()) <= 5e-6 * float(ref.abs().max())


def test_loftr_layer_golden(ops, mg):
    """the whole encoder layer assembled from kernels, against the reference layer output"""
    from cofii2p_amd.transformer import loftr_layer

    w = {k[len("lay_w_"):]: G(mg[k]) for k in mg.files if k.startswith("lay_w_")}
    out = loftr_layer(w, G(mg["lay_x"]), G(mg["lay_src"]))
    close(out, mg["lay_out"], 5e-5)


def test_small_glue(ops, mg):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1280, 128, generator=g)
    close(ops.col_inv_norm(G(x)), 1.0 / x.norm(dim=0).clamp_min(1e-12), 1e-5)
    close(ops.l2norm_rows(G(x)), torch.nn.functional.normalize(x, dim=1), 1e-6)
    close(ops.l2norm_rows(G(x), transpose=True), torch.nn.functional.normalize(x, dim=1).t(), 1e-6)
    close(ops.transpose(G(x)), x.t(), 0)
    close(ops.col_mean(G(x), 4), x.reshape(4, 320, 128).mean(1), 1e-6)
    out = torch.zeros(1280, 128, device=DEV)
    ops.pos_sine(G(mg["pe_grid"], torch.int32), out, accumulate=False)
    close(out, mg["pe_grid_out"], 1e-5)
    out = torch.ones(200, 128, device=DEV)
    ops.pos_sine(G(mg["pe_xyz"]), out, accumulate=True)
    close(out, mg["pe_xyz_out"] + 1.0, 2e-4)  # fp32 sin/cos of arguments up to ~500 rad


@pytest.mark.parametrize("S,Q,k", [(4096, 512, 128), (1000, 1000, 128), (2048, 300, 16), (10, 4, 16), (5000, 77, 1), (20480, 256, 128)])
def test_knn_bit_exact(ops, S, Q, k):
    from cofii2p_amd.synth import make_frame

    fr = make_frame(7, max(S, 64))
    sup = fr.points[:S].copy()
    if S >= 1000:
        sup[S // 2:] = sup[np.random.RandomState(0).choice(S // 2, S - S // 2)]  # exact duplicates -> ties
    qry = np.concatenate([sup[: Q // 2], sup[:Q - Q // 2] + np.float32(0.03)])
    ic, dc = knn_c.knn(sup, qry, k, True)
    ig, dg = ops.knn(G(sup), G(qry), k, return_dist=True)
    assert np.array_equal(ig.cpu().numpy().astype(np.int64), ic)
    assert np.array_equal(dg.cpu().numpy(), dc)
    assert np.array_equal(ops.nearest_node(G(sup), G(qry)).cpu().numpy().astype(np.int64), knn_c.nearest(sup, qry))


def _knn_cases():
    from cofii2p_amd.synth import make_frame

    rs = np.random.RandomState(5)
    kitti = make_frame(9, 20480).points
    cube = rs.uniform(-20, 20, (8192, 3)).astype(np.float32)                                   # no thin axis: cells are columns
    lattice = np.round(rs.uniform(-15, 15, (6000, 3)) * 2).astype(np.float32) / 2              # snapped: thousands of exact ties
    line = np.zeros((5000, 3), np.float32); line[:, 0] = rs.uniform(-100, 100, 5000)           # two degenerate axes
    same = np.tile(np.array([[3.0, -1.0, 7.0]], np.float32), (4200, 1))                        # all points identical: order by index
    cluster = np.concatenate([rs.normal(0, 0.3, (6000, 3)), rs.uniform(-500, 500, (300, 3))]).astype(np.float32)
    far_q = np.concatenate([kitti[:200], kitti[:200] + np.float32(400.0), kitti[:100] * np.float32(-3.0)])   # queries outside the box
    return [("kitti self", kitti, kitti, 128), ("kitti sub", kitti, kitti[rs.choice(20480, 10240)], 128), ("kitti far", kitti, far_q, 128),
            ("cube", cube, cube[:3000] + np.float32(0.01), 128), ("lattice", lattice, lattice, 128), ("line", line, line[:1500], 128),
            ("same", same, same[:300], 128), ("cluster", cluster, cluster[::3], 128), ("k1", kitti[:5000], kitti[:700], 1),
            ("k64", cube, cube[:999], 64), ("k65", cube, cube[:999], 65), ("tiny", kitti[:50], kitti[:64], 128),
            ("one", kitti[:1], kitti[:5], 16)]


def test_knn_grid_equals_brute_force(ops):
    """the cell-grid search returns the brute-force kernel's rows bit for bit — indices and distances — on every geometry"""
    for name, sup, qry, k in _knn_cases():
        S, Q = G(sup), G(qry)
        ib, db = ops.knn(S, Q, k, return_dist=True)
        grid = ops.KnnGrid(S)
        assert sorted(grid.order.cpu().tolist()) == list(range(sup.shape[0])), name        # cell order is a permutation
        ig, dg = ops.knn(S, Q, k, return_dist=True, grid=grid)
        assert torch.equal(ig, ib), name
        assert torch.equal(dg, db), name
        if qry.shape[0] == sup.shape[0]:                                                    # self search in cell order
            io, do = ops.knn(S, Q, k, return_dist=True, grid=grid, qorder=grid.order)
            assert torch.equal(io, ib) and torch.equal(do, db), name
    with pytest.raises(Exception):
        ops.knn(G(_knn_cases()[3][1]), G(_knn_cases()[3][2]), 16, grid=grid)               # grid of another support set


def test_knn_grid_random_geometries(ops):
    """randomised differential test of the pruning bound: anisotropic boxes, clusters, large offsets (where the canonical fp32
    distance is dominated by rounding and the bound has to widen the search), lattices; every row bit-equal to brute force"""
    rs = np.random.RandomState(77)
    for case in range(40):
        S = int(rs.choice([300, 1500, 4000, 9000]))
        Q = int(rs.choice([64, 500, 1500]))
        k = int(rs.choice([1, 7, 64, 100, 128]))
        scale = rs.uniform(0.05, 60.0, 3) * rs.choice([1.0, 1e-3, 1.0, 1.0], 3)         # thin or flat boxes now and then
        offset = rs.choice([0.0, 0.0, 50.0, 3000.0, -20000.0]) * rs.uniform(0.5, 1.0, 3)
        kind = case % 4
        if kind == 0:
            sup = rs.uniform(-1, 1, (S, 3))
        elif kind == 1:
            sup = rs.normal(0, 0.2, (S, 3)) + rs.choice([-1.0, 0.0, 1.0], (S, 1))
        elif kind == 2:
            sup = np.round(rs.uniform(-1, 1, (S, 3)) * 6) / 6
        else:
            sup = np.concatenate([rs.uniform(-1, 1, (S - S // 8, 3)), rs.normal(0, 1e-3, (S // 8, 3))])
        sup = (sup * scale + offset).astype(np.float32)
        qry = sup[rs.choice(S, Q)] + (rs.normal(0, 1, (Q, 3)) * scale * rs.choice([0.0, 0.01, 0.5, 3.0])).astype(np.float32)
        qry = qry.astype(np.float32)
        Sg, Qg = G(sup), G(qry)
        ib, db = ops.knn(Sg, Qg, k, return_dist=True)
        ig, dg = ops.knn(Sg, Qg, k, return_dist=True, grid=ops.KnnGrid(Sg))
        assert torch.equal(ig, ib) and torch.equal(dg, db), (case, S, Q, k, scale, offset)


def test_knn_grid_vs_c_oracle_and_pyramid(ops):
    """grid search against oracle/knn_oracle.c directly, and build_pyramid (grids on the large stages) against the brute-force pyramid"""
    from cofii2p_amd import preprocess
    from cofii2p_amd.synth import make_frame, subsample_indices

    pts = make_frame(13, 8192).points
    rows = np.random.RandomState(1).choice(8192, 400, replace=False)
    ic, dc = knn_c.knn(pts, pts[rows], 128, True)
    sup = G(pts)
    ig, dg = ops.knn(sup, G(pts[rows]), 128, return_dist=True, grid=ops.KnnGrid(sup))
    assert np.array_equal(ig.cpu().numpy().astype(np.int64), ic) and np.array_equal(dg.cpu().numpy(), dc)
    sub = [torch.from_numpy(s_).to(DEV) for s_ in subsample_indices(20480, 5, seed=3)]
    p0 = G(make_frame(14, 20480).points)
    got = preprocess.build_pyramid(p0, sub)
    saved, ops.KNN_GRID_MIN_SUPPORT = ops.KNN_GRID_MIN_SUPPORT, 1 << 30
    try:
        want = preprocess.build_pyramid(p0, sub)
    finally:
        ops.KNN_GRID_MIN_SUPPORT = saved
    for key in ("neighbors", "subsampling", "upsampling"):
        for a, b in zip(got[key], want[key]):
            assert torch.equal(a, b), key


def test_zero_row_inputs(ops):
    """zero query / activation rows are no-ops that return correctly shaped empty results; empty support sets are errors"""
    z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=DEV)
    sup = G(np.random.RandomState(0).normal(size=(2000, 3)).astype(np.float32))
    assert ops.knn(sup, z(0, 3), 16).shape == (0, 16)
    i0, d0 = ops.knn(sup, z(0, 3), 16, return_dist=True, grid=ops.KnnGrid(sup))
    assert i0.shape == (0, 16) and d0.shape == (0, 16) and i0.dtype == torch.int32
    assert ops.nearest_node(sup, z(0, 3)).shape == (0,)
    assert ops.gemm(z(0, 64), G(torch.randn(32, 64))).shape == (0, 32)
    assert ops.gather_rows(sup, z(0, dt=torch.int32)).shape == (0, 3)
    assert ops.neighbor_maxpool(G(torch.randn(10, 32)), z(0, 128, dt=torch.int32)).shape == (0, 32)
    assert ops.l2norm_rows(z(0, 64)).shape == (0, 64) and ops.l2norm_rows(z(0, 64), transpose=True).shape == (64, 0)
    assert ops.layer_norm(z(0, 64), G(torch.ones(64)), G(torch.zeros(64))).shape == (0, 64)
    cnt = torch.zeros(2, dtype=torch.int32, device=DEV)                       # a frame with no accepted match: kernels read count = 0
    fxy, best = ops.fine_match(z(8, 64, 16), z(8, 64), z(2, 8), cnt, 1.0)
    assert fxy.shape == (2, 8) and best.shape == (8,)
    for bad in (lambda: ops.knn(z(0, 3), sup[:5].contiguous(), 16), lambda: ops.KnnGrid(z(0, 3)), lambda: ops.nearest_node(z(0, 3), sup[:5].contiguous())):
        with pytest.raises(Exception):
            bad()
    torch.cuda.synchronize()


def test_idx_convert(ops):
    a = torch.randint(0, 20481, (1000, 128))
    assert torch.equal(ops.idx_to_int64(ops.idx_to_int32(G(a))).cpu(), a)


def test_matching_chain(ops, mg):
    pc, img, score = torch.from_numpy(mg["fp_pc"]), torch.from_numpy(mg["fp_img"]), torch.from_numpy(mg["fp_score"]).flatten()
    N = pc.shape[1]
    sim = ops.gemm(G(pc.t().contiguous()), G(img[0].reshape(128, -1).t().contiguous()))
    pix = ops.row_argmin_1m(sim)
    thr = np.array(O.score_thresholds(), dtype=np.float32)
    sel, xy, cnt = ops.select_matches(G(score), pix, 64, 20, thr)
    n = int(cnt[0])
    assert n == len(mg["fp_sel"]) and int(cnt[1]) == 0
    assert np.array_equal(sel[:n].cpu().numpy(), mg["fp_sel"])
    assert np.array_equal(xy[:, :n].cpu().numpy(), mg["fp_xy"])
    # threshold fallback: require more matches than the first threshold yields
    sel2, xy2, cnt2 = ops.select_matches(G(score), pix, 64, 20, thr, min_matches=n + 1)
    assert int(cnt2[0]) > n and int(cnt2[1]) > 0
    xy_ref, sel_ref = O.fine_process(score, pc, img[0], float(thr[int(cnt2[1])]))
    assert np.array_equal(sel2[: int(cnt2[0])].cpu().numpy(), sel_ref.numpy())
    # point2node
    assert np.array_equal(ops.nearest_node(G(mg["p2n_nodes"]), G(mg["p2n_pts"])).cpu().numpy(), mg["p2n_out"])
    # patches: val-style centres at scale 1
    ctr = G(mg["ep_ctr"])
    c12 = torch.tensor([12, 0], dtype=torch.int32, device=DEV)
    fmap = G(mg["ep_fmap"])   # (C, H2, W2) as the reference holds it; the product keeps maps pixel-major (NHWC)
    Cc, H2, W2 = fmap.shape
    pat = ops.extract_patches_nhwc(ops.transpose(fmap.reshape(Cc, H2 * W2)), H2, W2, ctr, c12, 12, 1.0)
    close(pat.reshape(12, 8, 4, 4), mg["ep_out"], 0)
    from cofii2p_amd.network import extract_patch

    close(extract_patch(fmap[None], ctr)[:, 0], mg["ep_out"], 0)   # the reference's free function, (B,C,H,W) in
    fxy, best = ops.fine_match(G(mg["fm_patches"]), G(mg["fm_pc"]), ctr, c12, 1.0)
    assert np.array_equal(best.cpu().numpy(), mg["fm_pred"])
    assert np.array_equal(fxy.cpu().numpy(), mg["fm_xy"])


def test_match_finish_equals_the_five_kernels(ops):
    """cofi_match_finish (coarse point + point2node + fine descriptor + 4x4 patch + fine matching in one launch) against the chain of
    the five stand-alone kernels: bit-identical, incl. patches over the image border and distance ties in the node search"""
    g = torch.Generator().manual_seed(11)
    N4, N1, C, H2, W2 = 160, 1500, 64, 40, 128
    pts1 = torch.round(torch.randn(N1, 3, generator=g) * 4) / 4          # a lattice: exact distance ties
    pts4 = pts1[torch.randperm(N1, generator=g)[:N4]] + torch.round(torch.randn(N4, 3, generator=g)) / 8
    fmap, fpc = torch.randn(H2 * W2, C, generator=g), torch.randn(N1, C, generator=g)
    n = 97
    sel = torch.full((N4,), 0, dtype=torch.int32)
    sel[:n] = torch.sort(torch.randperm(N4, generator=g)[:n]).values.int()
    xy = torch.zeros(2, N4)
    xy[0, :n] = torch.randint(0, W2 // 4, (n,), generator=g).float()     # coarse pixels incl. 0 and the last column / row: patches leave the map
    xy[1, :n] = torch.randint(0, H2 // 4, (n,), generator=g).float()
    cnt = torch.tensor([n, 0], dtype=torch.int32)
    a = [G(t) for t in (pts4, pts1, sel, cnt, fmap, xy, fpc)]
    cp, pat, fp, fxy, best = ops.match_finish(a[0], a[1], a[2], a[3], a[4], H2, W2, a[5], a[6], 4.0)
    cp0 = ops.gather_points_sel(a[0], a[2], a[3])
    node = ops.nearest_node_sel(a[1], a[0], a[2], a[3])
    pat0 = ops.extract_patches_nhwc(a[4], H2, W2, a[5], a[3], N4, 4.0)
    fp0 = ops.gather_rows_sel(a[6], node, a[3], N4)
    fxy0, best0 = ops.fine_match(pat0, fp0, a[5], a[3], 4.0)
    for got, want in ((cp, cp0), (pat, pat0), (fp, fp0), (best, best0)):
        assert torch.equal(got[:n], want[:n])
    assert torch.equal(fxy[:, :n], fxy0[:, :n])
    d = ((pts4[sel[:n].long()][:, None, :] - pts1[None]) ** 2).sum(-1)   # the node is a nearest one (lowest index among exact ties)
    assert torch.equal(fp[:n].cpu(), fpc[d.argmin(1)]) or float((d.gather(1, node[:n].cpu().long()[:, None])[:, 0] - d.min(1).values).abs().max()) < 1e-5


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16x6"])
def test_l2norm_epilogue(ops, monkeypatch, mode):
    """COFI_GEMM_L2NORM: F.normalize(dim=1) of the output rows in the GEMM / convolution epilogue (un-split and split-K plans, bias /
    residual / ReLU in front of it) against the stand-alone l2norm_rows of the same contraction; l2norm_rows2 = l2norm_rows twice"""
    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    g = torch.Generator().manual_seed(31)
    for M, N, K in [(10240, 64, 768), (1280, 128, 512), (200, 100, 60), (1280, 64, 7680)]:
        a, w, b = G(torch.randn(M, K, generator=g)), G(torch.randn(N, K, generator=g) / K ** 0.5), G(torch.randn(N, generator=g))
        close(ops.gemm(a, w, bias=b, l2norm=True), ops.l2norm_rows(ops.gemm(a, w, bias=b)), 2e-6)
    H, W, Cin, Cout = 40, 128, 64, 64
    x, wt = G(torch.randn(H * W, Cin, generator=g)), G(torch.randn(Cout, 9 * Cin, generator=g) / 24)
    b, res = G(torch.randn(Cout, generator=g)), G(torch.randn(H * W, Cout, generator=g))
    y = ops.conv2d_nhwc(x, H, W, wt, 3, bias=b, res=res, act=ops.ACT_RELU, l2norm=True)[0]
    close(y, ops.l2norm_rows(ops.conv2d_nhwc(x, H, W, wt, 3, bias=b, res=res, act=ops.ACT_RELU)[0]), 2e-6)
    with pytest.raises(Exception):
        ops.gemm(a, G(torch.randn(256, 7680, generator=g)), l2norm=True)   # a row must fit one tile
    o1, o2 = torch.empty((300, 96), device=DEV), torch.empty((300, 200), device=DEV)[:, 8:104]
    xs = G(torch.randn(300, 96, generator=g))
    ops.l2norm_rows2(xs, o1, o2)
    assert torch.equal(o1, ops.l2norm_rows(xs)) and torch.equal(o2, o1)


def test_bf16x6_presplit_weight_planes_are_bit_identical(ops, monkeypatch):
    """COFI_GEMM_BF16X6 | COFI_GEMM_W_SPLIT (three pre-split weight planes, opt-in ops.X6_W_SPLIT) = the on-the-fly split, bit for bit:
    dense, split-K, normalising loader and convolution launches"""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    g = torch.Generator().manual_seed(17)
    rn = lambda *s: torch.randn(*s, generator=g)
    for M, N, K in [(1280, 128, 128), (1280, 512, 7680), (300, 200, 36), (20480, 64, 576)]:
        a, w, b = G(rn(M, K)), ops.presplit(G(rn(N, K) / K ** 0.5)), G(rn(N))
        res = []
        for flag in (False, True):
            monkeypatch.setattr(ops, "X6_W_SPLIT", flag)
            res.append(ops.gemm_colstats(a, w, bias=b, act=ops.ACT_LEAKY01))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (M, N, K)
    H, W, Cin, Cout = 20, 64, 128, 128
    x, wt = G(rn(H * W, Cin)), ops.presplit(G(rn(Cout, 9 * Cin) / 34))
    y0, p0 = ops.gemm_colstats(G(rn(H * W, 64)), G(rn(Cin, 64) / 8))
    nx = ops.Normed(y0, ops.ColStats(p0, H * W, Cin), slope=0.0)
    res = []
    for flag in (False, True):
        monkeypatch.setattr(ops, "X6_W_SPLIT", flag)
        res.append((ops.conv2d_nhwc(x, H, W, wt, 3, colstats=True), ops.conv2d_nhwc(nx, H, W, wt, 3, colstats=True)))
    for i in range(2):
        assert torch.equal(res[0][i][0], res[1][i][0]) and torch.equal(res[0][i][1], res[1][i][1])


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16x6"])
def test_128_row_tiles_with_a_half_empty_last_tile(ops, monkeypatch, mode):
    """a 128 x 128 plan on M % 128 in (0, 64]: the last tile has ONE statistics slab - nothing may be written behind the table
    (round 4: the second half's zero sums went to slab nslab, past the allocation)"""
    import ctypes

    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    lib = ops._lib.load()
    force = lib.cofi_tune_force_plan
    force.argtypes, force.restype = [ctypes.c_int] * 3, ctypes.c_int
    g = torch.Generator().manual_seed(3)
    M, N, K = 320, 256, 128
    a, w = G(torch.randn(M, K, generator=g)), G(torch.randn(N, K, generator=g) / K ** 0.5)
    want, wpart = ops.gemm_colstats(a, w)
    big = torch.full((8, N, 2), 7.0, device=DEV)    # the (5, N, 2) table lives in front of 3 canary slabs
    try:
        assert force(128, 128, 1) == 0
        ws = ops._WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, N, K), a.device)
        wp, wld, wflag = ops._wargs(w)
        out = torch.empty((M, N), device=DEV)
        rc = lib.cofi_gemm_f32_fused(ops._p(a), K, None, wp, wld, ops._p(out), N, M, N, K, None, None, ops._gemm_flag() | wflag, ops._p(big), 1, ops._p(ws),
                                     0 if ws is None else ws.numel(), 1, ops._stream())
        assert rc == 0
    finally:
        force(0, 0, 0)
    torch.cuda.synchronize()
    assert bool((big[5:] == 7.0).all())
    close(out, want, 1e-5), close(big[:5], wpart, 1e-3)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1280, 512, 7680), (77, 33, 60), (2560, 1024, 3072), (1280, 128, 256), (20480, 32, 64),
                                   # small grids with >= 3 K-tiles per workgroup: the two-tiles-in-flight configuration (odd / even / ragged tile counts)
                                   (320, 256, 2304), (80, 512, 4608), (1280, 128, 1152), (100, 64, 1000), (64, 64, 384), (130, 60, 516),
                                   (2400, 256, 128), (4700, 512, 64), (1350, 1024, 256), (2381, 2048, 64)])  # XCD-contiguous tile order
def test_gemm_bf16x3_split(ops, M, N, K, monkeypatch):
    """3-term bf16 split with fp32 accumulation: error ~2^-16 per product, i.e. well inside the 1e-3 budget"""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x3")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 3
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    out_g = ops.gemm(G(a), G(w), bias=G(bias))
    out = out_g.cpu().double()
    scale = (a.double().abs() @ w.double().abs().t())  # sum of |products|: the natural error scale
    rel = ((out - ref).abs() / scale).max()
    assert rel < 3e-5, float(rel)
    # weights split once into bf16 planes: same rounding as the on-the-fly split -> bit-identical result
    if K % 4 == 0:
        out_s = ops.gemm(G(a), ops.presplit(G(w)), bias=G(bias))
        assert torch.equal(out_s, out_g)
    y, part = ops.gemm_colstats(G(a), G(w), bias=G(bias))
    st = ops.group_stats_from_colpart(part, M, N).cpu()  # one group per column
    assert float((st[:, 0].double() - ref.mean(0)).abs().max()) < 1e-4
    if N <= 128:
        ga, be = torch.ones(N), torch.zeros(N)
        ln = ops.gemm_layernorm(G(a), G(w), G(ga), G(be), bias=G(bias))
        close(ln, torch.nn.functional.layer_norm(ref.float(), (N,)), 2e-4)


@pytest.mark.parametrize("M,C", [(64, 64), (20480, 480), (1280, 128), (130, 68), (77, 33), (4096, 4), (8, 2048), (1000, 260)])
def test_transpose_shapes(ops, M, C):
    """cofi_transpose: the 16-byte tile kernel (everything a multiple of 4) and the scalar one, incl. a strided source view"""
    x = torch.arange(M * (C + 4), dtype=torch.float32).reshape(M, C + 4)
    xg = G(x)
    assert torch.equal(ops.transpose(xg[:, :C]).cpu(), x[:, :C].t())
    assert torch.equal(ops.transpose(xg.contiguous()[:, :C].contiguous()).cpu(), x[:, :C].t())


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1280, 512, 7680), (77, 33, 60), (2560, 1024, 3072), (20480, 32, 480), (100, 64, 1000), (64, 480, 20480),
                                   (4700, 512, 64), (1350, 1024, 256)])
def test_gemm_bf16x6_is_fp32_grade(ops, M, N, K, monkeypatch):
    """6-term bf16 split (three planes per operand): against fp64 the contraction is as accurate as the exact-fp32 MFMA kernel on the same
    operands - what the split drops is below 2^-24 of |a||b|, the rest is fp32 accumulation - on every tile configuration / split-K plan,
    with the fused epilogues (column statistics, LayerNorm) and the implicit-GEMM convolution."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 3
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.double().abs() @ w.double().abs().t())
    err = {}
    for mode in ("f32", "bf16x6", "bf16x3"):
        monkeypatch.setattr(ops, "GEMM_MODE", mode)
        out = ops.gemm(G(a), G(w), bias=G(bias)).cpu().double()
        err[mode] = (float(((out - ref).abs() / scale).max()), float((out - ref).pow(2).mean().sqrt()))
    assert err["bf16x6"][0] < max(2.0 * err["f32"][0], 2e-7) and err["bf16x6"][1] < 1.5 * err["f32"][1], err
    assert err["bf16x6"][1] < 0.3 * err["bf16x3"][1], err
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    y, part = ops.gemm_colstats(G(a), G(w), bias=G(bias))
    st = ops.group_stats_from_colpart(part, M, N).cpu()
    assert float((st[:, 0].double() - ref.mean(0)).abs().max()) < 1e-4
    assert torch.equal(y, ops.gemm(G(a), ops.presplit(G(w)), bias=G(bias)))   # a pre-split weight object hands its fp32 original to this arithmetic
    if N <= 128:
        ln = ops.gemm_layernorm(G(a), G(w), G(torch.ones(N)), G(torch.zeros(N)), bias=G(bias))
        close(ln, torch.nn.functional.layer_norm(ref.float(), (N,)), 2e-5)


def test_bf16x6_normalising_loader_and_convolution(ops, monkeypatch):
    """the pending-normalisation loader and the implicit-GEMM convolution in the 6-term arithmetic: fused == apply kernel + plain launch bit
    for bit, and the results at the exact-fp32 kernel's distance from torch's fp64 reference"""
    import torch.nn.functional as F

    from cofii2p_amd.image import _nhwc_weight

    g = torch.Generator().manual_seed(5)
    M, Kc, N, groups = 1280, 128, 512, 32
    x0, w0 = torch.randn(M, 96, generator=g) * 1.5 + 0.2, torch.randn(Kc, 96, generator=g) / 9.0
    w1, b1 = torch.randn(N, Kc, generator=g) / Kc ** 0.5, torch.randn(N, generator=g)
    ga, be = 1 + 0.2 * torch.randn(Kc, generator=g), 0.3 * torch.randn(Kc, generator=g)
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    y, part = ops.gemm_colstats(G(x0), ops.presplit(G(w0)), stat_width=4)
    nm = ops.Normed(y, ops.ColStats(part, M, groups, 1, width=4), G(ga), G(be), 0.1)
    assert nm.fusable()
    fused = ops.gemm(nm, ops.presplit(G(w1)), bias=G(b1))
    plain = ops.gemm(nm.materialize(), G(w1), bias=G(b1))
    assert torch.equal(fused, plain)
    ref = F.leaky_relu(F.group_norm((x0.double() @ w0.double().t()).t()[None], groups, ga.double(), be.double(), 1e-5)[0].t(), 0.1) @ w1.double().t() + b1.double()
    close(fused, ref.float(), 2e-5)
    Cin, Cout, H, W = 64, 128, 20, 32
    x = torch.randn(1, Cin, H, W, generator=g)
    wa = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    refc = F.conv2d(x.double(), wa.double(), stride=2, padding=1)
    errs = {}
    for mode in ("f32", "bf16x6"):
        monkeypatch.setattr(ops, "GEMM_MODE", mode)
        out = ops.conv2d_nhwc(G(_nhwc(x)), H, W, G(_nhwc_weight(wa)), 3, 2, 1)[0]
        errs[mode] = float((out.cpu().double() - _nhwc(refc)).abs().max())
    assert errs["bf16x6"] < max(2.0 * errs["f32"], 1e-6), errs


def _nhwc(t):  # (1,C,H,W) -> (H*W, C)
    return t[0].permute(1, 2, 0).reshape(-1, t.shape[1]).contiguous()


@pytest.mark.parametrize("Cin,Cout,H,W,ks,stride,pad", [(64, 64, 40, 128, 3, 1, 1), (64, 128, 40, 128, 3, 2, 1), (64, 128, 40, 128, 1, 2, 0),
                                                        (192, 128, 40, 128, 3, 1, 1), (256, 512, 10, 32, 3, 2, 1), (32, 16, 7, 9, 3, 1, 1)])
def test_conv2d_nhwc_implicit_gemm(ops, Cin, Cout, H, W, ks, stride, pad):
    import torch.nn.functional as F

    from cofii2p_amd.image import _nhwc_weight

    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    y, Ho, Wo = ops.conv2d_nhwc(G(_nhwc(x)), H, W, G(_nhwc_weight(w)), ks, stride, pad, bias=G(b))
    assert (Ho, Wo) == tuple(ref.shape[2:])
    close(y, _nhwc(ref), 2e-4)
    r = torch.randn(Ho * Wo, Cout, generator=g)
    y2, part, _, _ = ops.conv2d_nhwc(G(_nhwc(x)), H, W, G(_nhwc_weight(w)), ks, stride, pad, bias=G(b), res=G(r), act=ops.ACT_RELU, colstats=True)
    refr = torch.relu(_nhwc(ref) + r)
    close(y2, refr, 2e-4)
    st = ops.group_stats_from_colpart(part, Ho * Wo, Cout).cpu()
    close(st[:, 0], refr.mean(0), 2e-4)


def test_image_nhwc_glue(ops):
    import torch.nn.functional as F

    from cofii2p_amd.image import _nhwc_weight

    g = torch.Generator().manual_seed(4)
    img = torch.rand(1, 3, 64, 96, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    col, Ho, Wo = ops.im2col_stem(G(img[0]))
    y = ops.gemm(col, G(_nhwc_weight(w, 160)))
    close(y, _nhwc(F.conv2d(img, w, stride=2, padding=3)), 2e-4)
    x = torch.randn(1, 64, 32, 48, generator=g)
    yp, H2, W2 = ops.maxpool3x3s2_nhwc(G(_nhwc(x)), 32, 48)
    close(yp, _nhwc(F.max_pool2d(x, 3, 2, 1)), 0)
    low, skip = torch.randn(1, 128, 20, 64, generator=g), torch.randn(1, 64, 40, 128, generator=g)
    ref = torch.cat([F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=False), skip], 1)
    close(ops.upsample2x_cat_nhwc(G(_nhwc(low)), 20, 64, G(_nhwc(skip))), _nhwc(ref), 1e-6)
    # InstanceNorm + ReLU + residual through the per-channel GroupNorm machinery
    xm = _nhwc(x)
    yv, part = ops.gemm_colstats(G(xm), G(torch.eye(64)))
    out = ops.group_norm_apply(yv, ops.group_stats_from_colpart(part, xm.shape[0], 64), slope=0.0, res=G(xm))
    close(out, _nhwc(F.relu(F.instance_norm(x) + x)), 2e-5)


@pytest.mark.parametrize("frames,Mf,Kc,N,groups,affine,slope", [
    (1, 1280, 128, 512, 32, True, 0.1),      # unary2 of a deep block: K = mid
    (1, 20480, 32, 128, 32, True, 0.1),      # encoder1_2: 320 slabs, group width 1
    (1, 5120, 64, 256, 32, True, 0.1),
    (1, 1000, 256, 96, 32, True, 0.1),       # ragged rows (last slab / tile partial)
    (1, 1280, 128, 64, 128, False, 0.0),     # score head: InstanceNorm (one group per column) + ReLU
    (1, 2560, 512, 2048, 32, True, 0.1),     # K = 512 (largest table), 64-wide statistics groups downstream
    (3, 1280, 128, 128, 32, True, 0.1),      # stack mode: per-frame statistics
    (2, 640, 64, 1, 64, False, 0.0),         # N = 1 (last score layer)
])
def test_gemm_normalising_loader(ops, monkeypatch, frames, Mf, Kc, N, groups, affine, slope):
    """cofi_gemm_f32_fused with a pending GroupNorm / InstanceNorm on A == stand-alone apply kernel followed by the same GEMM,
    bit for bit (same operations in the same order), and the apply kernel itself against torch's group_norm"""
    import torch.nn.functional as F

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x3")
    g = torch.Generator().manual_seed(Mf + Kc + N)
    M = frames * Mf
    x0 = torch.randn(M, 96, generator=g) * 1.5 + 0.2
    w0 = torch.randn(Kc, 96, generator=g) / 9.0
    w1 = torch.randn(N, Kc, generator=g) / Kc ** 0.5
    b1 = torch.randn(N, generator=g)
    ga = (1 + 0.2 * torch.randn(Kc, generator=g)) if affine else None
    be = (0.3 * torch.randn(Kc, generator=g)) if affine else None
    cpg = Kc // groups
    for sw in sorted({1, cpg}):
        y, part = ops.gemm_colstats(G(x0), ops.presplit(G(w0)), stat_width=sw)
        assert part.shape == (frames * ((Mf + 63) // 64), Kc // sw, 2)
        st = ops.ColStats(part, M, groups, frames, width=sw)
        assert st.fusable()
        nm = ops.Normed(y, st, None if ga is None else G(ga), None if be is None else G(be), slope)
        mat = nm.materialize()
        yc = y.cpu()
        ref = torch.cat([F.leaky_relu(F.group_norm(yc[f * Mf:(f + 1) * Mf].t()[None], groups, ga, be, 1e-5)[0].t(), slope) for f in range(frames)])
        close(mat, ref, 3e-5)
        w1s = ops.presplit(G(w1))
        fused, fpart = ops.gemm_colstats(nm, w1s, bias=G(b1), frames=frames)
        plain, ppart = ops.gemm_colstats(mat, w1s, bias=G(b1))
        assert torch.equal(fused, plain)
        assert torch.equal(fpart, ppart)
        close(plain, (ref.double() @ w1.double().t() + b1.double()).float(), 2e-4)
    if cpg > 1:   # the group-wide table is the per-column table summed over the group's columns
        _, p1 = ops.gemm_colstats(G(x0), ops.presplit(G(w0)), stat_width=1)
        close(part, p1.reshape(p1.shape[0], Kc // cpg, cpg, 2).sum(2), 2e-5 * float(p1.abs().max()))


@pytest.mark.parametrize("frames,Cin,Cout,H,W,stride", [(1, 64, 64, 40, 128, 1), (1, 128, 128, 20, 64, 1), (1, 256, 256, 10, 32, 1), (2, 64, 64, 16, 16, 1),
                                                        (1, 64, 128, 40, 128, 2)])
def test_conv2d_normalising_loader(ops, monkeypatch, frames, Cin, Cout, H, W, stride):
    """second convolution of a BasicBlock (imagenet.py:58-66): InstanceNorm + ReLU of the first convolution's output applied by the
    implicit-GEMM loader (zero padding AFTER the normalisation) == apply kernel + plain convolution, and against torch"""
    import torch.nn.functional as F

    from cofii2p_amd.image import _nhwc_weight

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x3")
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(frames, Cin, H, W, generator=g)
    wa = torch.randn(Cin, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    wb = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xm = torch.cat([_nhwc(x[f:f + 1]) for f in range(frames)])
    y1, part1, _, _ = ops.conv2d_nhwc(G(xm), H, W, ops.presplit(G(_nhwc_weight(wa))), 3, 1, 1, colstats=True, frames=frames)
    nm = ops.Normed(y1, ops.ColStats(part1, y1.shape[0], Cin, frames), slope=0.0)
    wbs = ops.presplit(G(_nhwc_weight(wb)))
    fused, fp, Ho, Wo = ops.conv2d_nhwc(nm, H, W, wbs, 3, stride, 1, colstats=True, frames=frames)
    plain, pp, _, _ = ops.conv2d_nhwc(nm.materialize(), H, W, wbs, 3, stride, 1, colstats=True, frames=frames)
    assert torch.equal(fused, plain) and torch.equal(fp, pp)
    ref = F.conv2d(F.relu(F.instance_norm(F.conv2d(x, wa, padding=1))), wb, stride=stride, padding=1)
    close(fused, torch.cat([_nhwc(ref[f:f + 1]) for f in range(frames)]), 3e-4)


def test_statistics_slabs_must_not_straddle_frames(ops):
    """2 frames of 112 rows give 4 slabs of 64 rows, divisible by the frame count although slab 1 mixes both frames:
    the per-frame folds must refuse such tables instead of silently mixing rows"""
    from cofii2p_amd._lib import CofiError

    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(224, 64, generator=g), torch.randn(128, 64, generator=g)
    y, part = ops.gemm_colstats(G(x), G(w))
    assert part.shape[0] == 4
    st = ops.ColStats(part, 224, 32, frames=2)
    assert not st.fusable()
    with pytest.raises(CofiError):
        st.finalize()
    with pytest.raises(CofiError):
        ops.col_inv_norm_from_colpart(part, 224, 128, frames=2)
    with pytest.raises(CofiError):
        ops.attention(y[:, :128], y[:, :128], y[:, :128], q_colpart=part, frames=2)
    # ... while whole-slab frames are served
    x2 = torch.randn(256, 64, generator=g)
    y2, part2 = ops.gemm_colstats(G(x2), G(w))
    assert ops.ColStats(part2, 256, 32, frames=2).fusable()
    st2 = ops.group_stats_from_colpart(part2, 256, 32, frames=2).cpu()
    ref = y2.cpu().reshape(2, 128, 32, 4)
    close(st2[:, :, 0], ref.mean((1, 3)), 1e-4)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("bf16x6", 2e-5)])
def test_loftr_layer_fused_tail(ops, mg, monkeypatch, mode, tol):
    """the one-kernel layer tail (merge+LN1+MLP+LN2+residual, cofi_loftr_tail) against the reference layer output, in both bf16-split
    arithmetics (2 / 3 planes per operand)"""
    from cofii2p_amd.transformer import loftr_layer

    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    w = {k[len("lay_w_"):]: G(mg[k]) for k in mg.files if k.startswith("lay_w_")}
    out = loftr_layer(w, G(mg["lay_x"]), G(mg["lay_src"]))  # L = 48: exercises the row tail of the 32-row tiles
    close(out, mg["lay_out"], tol)
    g = torch.Generator().manual_seed(8)
    x, src = torch.randn(1280, 128, generator=g), torch.randn(300, 128, generator=g)
    sd = {k: v.cpu() for k, v in w.items()}
    close(loftr_layer(w, G(x), G(src)), O.loftr_layer({"l." + k: v for k, v in sd.items()}, "l.", x, src), tol)


@pytest.mark.parametrize("frag", [True, False])
@pytest.mark.parametrize("mode,tol", [("bf16x3", 2e-4), ("bf16x6", 2e-5)])
def test_loftr_tail_fused_successors(ops, monkeypatch, mode, tol, frag):
    """cofi_loftr_tail with its optional successors: two projection segments of `out` (+ the 32-row column partials the attention
    kernel folds into the token-axis Q norm) and F.normalize(out, dim=1) token- and channel-major, against the same launch without
    them + fp64 torch; rows = 200: a partial last tile"""
    from cofii2p_amd.transformer import pack_layer

    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    monkeypatch.setattr(ops, "TAIL_FRAG", frag)   # weight planes in fragment order (default) / row-major: identical bits
    g = torch.Generator().manual_seed(21)
    rn = lambda *s: torch.randn(*s, generator=g)
    sd = {"q_proj.weight": rn(128, 128) / 11, "k_proj.weight": rn(128, 128) / 11, "v_proj.weight": rn(128, 128) / 11, "merge.weight": rn(128, 128) / 11,
          "mlp.0.weight": rn(256, 256) / 16, "mlp.2.weight": rn(128, 256) / 16, "norm1.weight": 1 + 0.1 * rn(128), "norm1.bias": 0.1 * rn(128),
          "norm2.weight": 1 + 0.1 * rn(128), "norm2.bias": 0.1 * rn(128)}
    w = pack_layer({k: G(v) for k, v in sd.items()}, "")
    for rows in (200, 256):
        msg, x = G(rn(rows, 128)), G(rn(rows, 128))
        base = ops.loftr_tail(msg, x, w, torch.empty_like(x))
        with monkeypatch.context() as mp:
            mp.setattr(ops, "TAIL_FRAG", not frag)
            assert torch.equal(base, ops.loftr_tail(msg, x, w, torch.empty_like(x)))   # the other weight layout: same bits
        sfx = ops.tail_suffix()
        y0, y1 = torch.empty((rows, 256), device=DEV), torch.empty((rows, 400), device=DEV)[:, 8:392]   # a strided destination
        part = torch.empty(((rows + 31) // 32, 384, 2), device=DEV) if rows % 32 == 0 else None
        l2, l2t = torch.empty((rows, 128), device=DEV), torch.empty((128, rows), device=DEV)
        out = ops.loftr_tail(msg, x, w, torch.empty_like(x), proj=[(w["kv" + sfx], y0, None), (w["qkv" + sfx], y1, part)], out_l2=l2, out_l2t=l2t)
        assert torch.equal(out, base)
        o64 = out.double().cpu()
        close(y0, (o64 @ torch.cat([sd["k_proj.weight"], sd["v_proj.weight"]]).double().t()).float(), tol)
        ref1 = o64 @ torch.cat([sd["q_proj.weight"], sd["k_proj.weight"], sd["v_proj.weight"]]).double().t()
        close(y1, ref1.float(), tol)
        assert torch.equal(l2, ops.l2norm_rows(out)) and torch.equal(l2t, l2.t())
        if part is not None:
            yb = y1.double().cpu().reshape(rows // 32, 32, 384)
            close(part[:, :, 0], yb.sum(1).float(), 1e-4)
            close(part[:, :, 1], (yb * yb).sum(1).float(), 1e-4)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 3e-4), ("bf16x6", 3e-5)])
@pytest.mark.parametrize("frames,L", [(1, 128), (2, 192)])
def test_transformer_fused_chain(ops, monkeypatch, mode, tol, frames, L):
    """the whole I2P transformer (transformer.py:85-104) as the fused chain - every tail computes the q / k / v of the layers that
    follow, the last tails the normalised descriptors - against the CPU oracle and against the layer-by-layer form"""
    from cofii2p_amd import transformer as T
    from cofii2p_amd.spec import LAYER_KINDS, N_LAYERS

    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    g = torch.Generator().manual_seed(5 + L)
    rn = lambda *s: torch.randn(*s, generator=g)
    sd = {}
    for l in range(N_LAYERS):
        p = "transformer.layers.%d." % l
        for n in ("q_proj", "k_proj", "v_proj", "merge"):
            sd[p + n + ".weight"] = rn(128, 128) / 11
        sd[p + "mlp.0.weight"], sd[p + "mlp.2.weight"] = rn(256, 256) / 16, rn(128, 256) / 16
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1 + 0.1 * rn(128), 0.1 * rn(128)
    layers = [T.pack_layer({k: G(v) for k, v in sd.items()}, "transformer.layers.%d." % l) for l in range(N_LAYERS)]
    xi, xp = rn(frames * L, 128), rn(frames * L, 128)

    def run(chain):
        monkeypatch.setattr(T, "FUSED_CHAIN", chain)
        ts = T.TokenStreams(frames * L, frames * L, 128, torch.device(DEV))
        ts.img[0][:, :128].copy_(G(xi)), ts.pc[0][:, :128].copy_(G(xp))
        l2 = (torch.empty((frames * L, 128), device=DEV), torch.empty((frames * L, 128), device=DEV),
              torch.empty((128, L), device=DEV) if frames == 1 else None, torch.empty((128, L), device=DEV) if frames == 1 else None)
        ti, tp, done = T.run_transformer(layers, LAYER_KINDS, ts, 4, frames=frames, l2=l2)
        assert done == chain
        return ti.clone(), tp.clone(), l2

    ci, cp, l2 = run(True)
    pi, pp, _ = run(False)
    close(ci, pi, tol), close(cp, pp, tol)
    for f in range(frames):
        ri, rp = O.transformer(sd, xi[f * L:(f + 1) * L], xp[f * L:(f + 1) * L])
        close(ci[f * L:(f + 1) * L], ri, tol), close(cp[f * L:(f + 1) * L], rp, tol)
    assert torch.equal(l2[0], ops.l2norm_rows(ci)) and torch.equal(l2[1], ops.l2norm_rows(cp))
    if frames == 1:
        assert torch.equal(l2[2], l2[0].t()) and torch.equal(l2[3], l2[1].t())


def test_workspace_growth_keeps_captured_addresses(ops):
    """a hipGraph captured on a slot records the scratch address; a later, larger request on the same slot must not free it"""
    ws = ops.Workspace()
    small = ws.get(1000, torch.device(DEV))
    addr = small.data_ptr()
    del small
    big = ws.get(50_000_000, torch.device(DEV))
    assert big.numel() >= 50_000_000 and big.data_ptr() != addr
    assert any(r.data_ptr() == addr for r in ws.retired)
    assert ws.get(2000, torch.device(DEV)).data_ptr() == big.data_ptr()   # grow-only: later small requests reuse the big buffer


def test_multi_copy(ops):
    """batched device-to-device copy: mixed dtypes, odd byte counts, unaligned views, cached descriptor tables"""
    g = torch.Generator().manual_seed(5)
    srcs = [G(torch.randn(1000, 3, generator=g)), G(torch.randint(0, 100, (77, 128), generator=g, dtype=torch.int32)),
            G(torch.randint(0, 255, (13,), generator=g, dtype=torch.uint8)), G(torch.randn(4097, generator=g))[1:], None]
    dsts = [torch.empty_like(s) if s is not None else None for s in srcs]
    mc = ops.MultiCopy(torch.device(DEV))
    for _ in range(2):  # second round: cached table
        for d in dsts:
            if d is not None:
                d.zero_()
        mc.run(srcs, dsts)
        for s_, d in zip(srcs, dsts):
            if s_ is not None:
                assert torch.equal(s_, d)
    assert len(mc.tables) == 1


def test_attention_stress_size_sampled_rows(attn):
    """BASELINE configs[4] (896x1600 image: 22 400 tokens): the L x S score matrix (8 GB per call in the reference) cannot be
    checked whole; a random sample of query rows is compared with an exact fp64 softmax of those rows"""
    L = S = 22400
    g = torch.Generator().manual_seed(77)
    q, k, v = torch.randn(L, 128, generator=g), torch.randn(S, 128, generator=g), torch.randn(S, 128, generator=g)
    out = attn.attention(G(q), G(k), G(v)).cpu()
    rows = torch.randperm(L, generator=g)[:192]
    qs = q[rows].double().view(-1, 4, 32)
    kd, vd = k.double().view(S, 4, 32), v.double().view(S, 4, 32)
    att = torch.softmax(torch.einsum("lhd,shd->hls", qs, kd) / 32 ** 0.5, dim=-1)
    ref = torch.einsum("hls,shd->lhd", att, vd).reshape(-1, 128)
    close(out[rows], ref.float(), 5e-5)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("name", ["self", "down", "up", "tiny"])
def test_knn_against_reference_knn(ops, name):
    """both HIP searches against the rows the reference's own knn() returned (tests/golden/knn_ref.npz): distances bit for bit,
    indices up to the ties torch.topk leaves unspecified - and bit-equal to the tie-defined C oracle"""
    from test_oracle_golden import _knn_case, check_knn_against_reference

    gold = load_golden("knn_ref.npz")
    support, query, k = _knn_case(gold, name)
    ic, dc = knn_c.knn(support, query, k, True)
    sup = G(support)
    for grid in (None, ops.KnnGrid(sup)):
        idx, dist = ops.knn(sup, G(query), k, return_dist=True, grid=grid)
        i, d = idx.cpu().numpy().astype(np.int64), dist.cpu().numpy()
        assert np.array_equal(i, ic) and np.array_equal(d, dc)
        check_knn_against_reference(i, d, gold["idx_" + name], gold["dist_" + name], support, query)


def test_knn_stress_size_properties(ops):
    """40 960 points, k = 128 (BASELINE configs[4]): sorted ascending, self first, and a sample of rows bit-equal to the oracle"""
    from cofii2p_amd.synth import make_frame

    pts = make_frame(11, 40960).points
    idx, dist = ops.knn(G(pts), G(pts), 128, return_dist=True)
    d = dist.cpu().numpy()
    assert (np.diff(d, axis=1) >= 0).all()                                   # sortedness
    i = idx.cpu().numpy().astype(np.int64)
    assert (d[:, 0] == d.min(1)).all() and (np.sort(i, 1)[:, 1:] != np.sort(i, 1)[:, :-1]).all()   # no duplicates in a row
    rows = np.random.RandomState(0).choice(40960, 256, replace=False)
    ic, dc = knn_c.knn(pts, pts[rows], 128, True)
    assert np.array_equal(i[rows], ic) and np.array_equal(d[rows], dc)


@pytest.mark.parametrize("M,N,K,frames", [(81920, 64, 96, 4), (65536 + 77, 64, 60, 1), (131072, 32, 128, 8), (66560, 48, 160, 4)])
def test_tall_tiles_for_narrow_outputs_bf16x6(ops, monkeypatch, M, N, K, frames):
    """bf16x6, <= 64 output columns over >= 512 tiles of 128 rows (stack-mode batches): make_plan takes 128 x 64 tiles
    (gemm.hip `make_plan`, measured in DESIGN.md section 12.2).  Same products in the same K order as the 64 x 64 tile: output, column
    statistics and the normalising loader's result must be BIT-equal to the forced 64 x 64 plan's, ragged last tiles and
    per-frame statistics included; and right against fp64."""
    import ctypes

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    lib = ops._lib.load()
    force = lib.cofi_tune_force_plan
    force.argtypes, force.restype = [ctypes.c_int] * 3, ctypes.c_int
    g = torch.Generator().manual_seed(M + N)
    a = G(torch.randn(M, K, generator=g))
    w = G(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = G(torch.randn(N, generator=g))
    w2 = ops.presplit(G(torch.randn(40, N, generator=g) / N ** 0.5))
    gam, bet = G(torch.randn(N, generator=g)), G(torch.randn(N, generator=g))
    groups = N // 2

    def run():
        y, part = ops.gemm_colstats(a, w, bias=bias, frames=frames) if M % frames == 0 and (M // frames) % 64 == 0 else ops.gemm_colstats(a, w, bias=bias)
        fr = frames if M % frames == 0 and (M // frames) % 64 == 0 else 1
        st = ops.ColStats(part, M, groups, fr)
        z = ops.gemm(ops.Normed(y, st, gam, bet, 0.1), w2, frames=fr) if st.fusable() else None
        return y.clone(), part.clone(), None if z is None else z.clone()

    tall = run()
    try:
        assert force(64, 64, 1) == 0
        flat = run()
    finally:
        force(0, 0, 0)
    for t, f in zip(tall, flat):
        assert (t is None) == (f is None)
        if t is not None:
            assert torch.equal(t, f)
    ref = a.double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    scale = a.double().cpu().abs() @ w.double().cpu().abs().t() + bias.double().cpu().abs()
    assert float(((tall[0].double().cpu() - ref).abs() / scale).max()) < 2e-6


def test_tall_tiles_convolution_bf16x6(ops, monkeypatch):
    """the 3 x 3 convolutions with 64 output channels of a stack-mode batch (implicit GEMM, 128 x 64 tiles): bit-equal to the 64 x 64 plan,
    pending InstanceNorm + ReLU of the producer applied by the loader included"""
    import ctypes

    from cofii2p_amd.image import _nhwc_weight

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    lib = ops._lib.load()
    force = lib.cofi_tune_force_plan
    force.argtypes, force.restype = [ctypes.c_int] * 3, ctypes.c_int
    frames, Cin, Cout, H, W = 4, 64, 64, 80, 256
    g = torch.Generator().manual_seed(11)
    x = G(torch.randn(frames * H * W, Cin, generator=g))
    wa = ops.presplit(G(_nhwc_weight(torch.randn(Cin, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)))
    wb = ops.presplit(G(_nhwc_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)))

    def run():
        y1, p1, _, _ = ops.conv2d_nhwc(x, H, W, wa, 3, 1, 1, colstats=True, frames=frames)
        nm = ops.Normed(y1, ops.ColStats(p1, y1.shape[0], Cin, frames), slope=0.0)
        y2, p2, _, _ = ops.conv2d_nhwc(nm, H, W, wb, 3, 1, 1, colstats=True, frames=frames)
        return y1.clone(), p1.clone(), y2.clone(), p2.clone()

    direct = lib.cofi_tune_force_conv_direct   # this test is about the two implicit-GEMM tilings: keep the direct 3 x 3 kernel out of it
    direct.argtypes, direct.restype = [ctypes.c_int], ctypes.c_int
    try:
        assert direct(-1) == 0
        tall = run()
        assert force(64, 64, 1) == 0
        flat = run()
    finally:
        force(0, 0, 0)
        direct(0)
    for t, f in zip(tall, flat):
        assert torch.equal(t, f)
    ref = torch.nn.functional.conv2d(x.cpu().reshape(frames, H, W, Cin).permute(0, 3, 1, 2), wa.w.cpu().reshape(Cin, 3, 3, Cin).permute(0, 3, 1, 2), padding=1)
    close(tall[0], ref.permute(0, 2, 3, 1).reshape(-1, Cin), 2e-5)


def _force_hooks(ops):
    import ctypes

    lib = ops._lib.load()
    fp, fb = lib.cofi_tune_force_plan, lib.cofi_tune_force_big
    fp.argtypes, fp.restype = [ctypes.c_int] * 3, ctypes.c_int
    fb.argtypes, fb.restype = [ctypes.c_int] * 2, ctypes.c_int
    return fp, fb


@pytest.mark.parametrize("M,N,K,frames,ks", [(4096, 256, 512, 1, 1), (4096 + 77, 256, 1024, 1, 1), (8192, 192, 512, 2, 1), (2048, 512, 2560, 2, 2),
                                            (1024, 128, 7680, 1, 3), (333, 130, 96, 1, 1), (20480, 512, 1536, 4, 1),
                                            # one / two K-tiles: the loop's prologue and its first, MFMA-less iteration alone
                                            (512, 128, 32, 1, 1), (768, 256, 64, 2, 1), (256, 128, 128, 1, 1)])
def test_big_tiles_bf16x6(ops, monkeypatch, M, N, K, frames, ks):
    """The 256 x 128 bf16x6 kernel (csrc/gemm_x6_big.inc: one workgroup per CU, split of tile t + 1 behind the MFMAs of tile t): same
    products in the same K order as the small-tile kernel, so at an equal K split output, column statistics and the normalising loader's
    result are BIT-equal to the forced 128 x 128 plan's - ragged last tiles in M and N, per-frame statistics, split-K included; and right
    against fp64."""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    monkeypatch.setattr(ops, "F16X3_BIG", False)   # this test is about the six-product kernel (the three-product one: test_big_tiles_f16x3)
    fp, fb = _force_hooks(ops)
    g = torch.Generator().manual_seed(M + N + K)
    a = G(torch.randn(M, K, generator=g))
    w = G(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = G(torch.randn(N, generator=g))
    N2 = 256
    w2 = ops.presplit(G(torch.randn(N2, N, generator=g) / N ** 0.5))
    gam, bet = G(torch.randn(N, generator=g)), G(torch.randn(N, generator=g))
    groups = 32 if N % 32 == 0 else 2
    per_frame = M % frames == 0 and (M // frames) % 256 == 0
    fr = frames if per_frame else 1

    def run():
        y, part = ops.gemm_colstats(a, w, bias=bias, frames=fr)
        st = ops.ColStats(part, M, groups, fr)
        z = ops.gemm(ops.Normed(y, st, gam, bet, 0.1), w2, frames=fr) if (st.fusable() and N % 32 == 0) else None
        return y.clone(), part.clone(), None if z is None else z.clone()

    try:
        assert fb(1, ks) == 0
        big = run()
        assert fb(-1, 0) == 0 and fp(128, 128, ks) == 0
        small = run()
    finally:
        fp(0, 0, 0)
        fb(0, 0)
    for t, f in zip(big, small):
        assert (t is None) == (f is None)
        if t is not None:
            assert torch.equal(t, f)
    ref = a.double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    scale = a.double().cpu().abs() @ w.double().cpu().abs().t() + bias.double().cpu().abs()
    assert float(((big[0].double().cpu() - ref).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize("pad,cmid", [(1, 128), (0, 128)])
def test_big_tiles_convolution_bf16x6(ops, monkeypatch, pad, cmid):
    """3 x 3 / stride 1 convolutions on the 256 x 128 kernel (implicit GEMM whose K-tiles lie inside one tap: per-row address register +
    scalar tap offset, validity mask per row): bit-equal to the small-tile plan, pending InstanceNorm + ReLU of the producer applied by
    the loader included, stack mode (frames), and right against torch's convolution."""
    from cofii2p_amd.image import _nhwc_weight

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    monkeypatch.setattr(ops, "F16X3_BIG", False)
    fp, fb = _force_hooks(ops)
    frames, Cin, Cmid, Cout, H, W = 2, 64, cmid, 256, 40 + 2 * (1 - pad), 128 + 2 * (1 - pad)
    g = torch.Generator().manual_seed(12)
    x = G(torch.randn(frames * H * W, Cin, generator=g))
    wa = ops.presplit(G(_nhwc_weight(torch.randn(Cmid, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)))
    wb = ops.presplit(G(_nhwc_weight(torch.randn(Cout, Cmid, 3, 3, generator=g) / (Cmid * 9) ** 0.5)))

    def run():
        y1, p1, Ho, Wo = ops.conv2d_nhwc(x, H, W, wa, 3, 1, pad, colstats=True, frames=frames)
        if pad == 1:
            nm = ops.Normed(y1, ops.ColStats(p1, y1.shape[0], Cmid, frames), slope=0.0)
            y2, p2, _, _ = ops.conv2d_nhwc(nm, Ho, Wo, wb, 3, 1, 1, colstats=True, frames=frames)
            return y1.clone(), p1.clone(), y2.clone(), p2.clone()
        return y1.clone(), p1.clone()

    try:
        assert fb(1, 1) == 0
        big = run()
        assert fb(-1, 0) == 0 and fp(128, 128, 1) == 0
        small = run()
    finally:
        fp(0, 0, 0)
        fb(0, 0)
    for t, f in zip(big, small):
        assert torch.equal(t, f)
    ref = torch.nn.functional.conv2d(x.cpu().reshape(frames, H, W, Cin).permute(0, 3, 1, 2), wa.w.cpu().reshape(Cmid, 3, 3, Cin).permute(0, 3, 1, 2), padding=pad)
    close(big[0], ref.permute(0, 2, 3, 1).reshape(-1, Cmid), 2e-5)


def test_pending_norm_with_groups_wider_than_the_finalize_kernel(ops, monkeypatch):
    """ColStats.desc: cofi_norm_finalize serves groups of <= 64 table columns; a wider group (256 channels in 2 groups) must fall back to the
    consumers' own fold of the partials instead of raising COFI_EUNSUPPORTED - same result as the materialised normalisation"""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    g = torch.Generator().manual_seed(5)
    M, C, N2 = 4096, 256, 128
    a = G(torch.randn(M, 64, generator=g))
    w = G(torch.randn(C, 64, generator=g) / 8)
    w2 = ops.presplit(G(torch.randn(N2, C, generator=g) / 16))
    gam, bet = G(torch.randn(C, generator=g)), G(torch.randn(C, generator=g))
    y, part = ops.gemm_colstats(a, w)
    st = ops.ColStats(part, M, 2, 1)
    d = st.desc(gam, bet, 0.1)
    assert not d.scale_shift            # no finalize launch for this table
    nm = ops.Normed(y, st, gam, bet, 0.1)
    z = ops.gemm(nm, w2)
    z_ref = ops.gemm(nm.materialize(), w2)
    close(z, z_ref, 1e-5)


@pytest.mark.parametrize("H,W,Cin,frames,res_act,force", [(8, 64, 64, 2, False, 1), (40, 128, 64, 2, True, 1), (12, 192, 32, 1, False, 1),
                                                          (8, 64, 64, 2, True, 2), (40, 128, 64, 2, False, 2), (80, 256, 64, 1, True, 2)])
def test_direct_3x3_convolution_bf16x6(ops, monkeypatch, H, W, Cin, frames, res_act, force):
    """the direct 3 x 3 / stride 1 / pad 1 kernel for 64 output channels (csrc/conv_direct.inc: the input halo of a 4 x 64-pixel tile split once
    into LDS planes, nine taps read it at shifted offsets) against the implicit-GEMM plan it replaces - other summation order of the same
    products: fp32-grade agreement, not bit equality -, against torch's convolution, with the producer's pending InstanceNorm + ReLU applied by
    the halo loader, column statistics, bias / residual / ReLU and stack mode; force = 1: 2-row tiles (small grids), 2: 4-row tiles."""
    import ctypes

    from cofii2p_amd.image import _nhwc_weight

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    hook = ops._lib.load().cofi_tune_force_conv_direct
    hook.argtypes, hook.restype = [ctypes.c_int], ctypes.c_int
    Cout = 64
    g = torch.Generator().manual_seed(H + W)
    x = G(torch.randn(frames * H * W, Cin, generator=g))
    wa = ops.presplit(G(_nhwc_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)))
    wb = ops.presplit(G(_nhwc_weight(torch.randn(Cout, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5)))
    bias = G(torch.randn(Cout, generator=g))
    res = G(torch.randn(frames * H * W, Cout, generator=g))

    def run():
        y1, p1, _, _ = ops.conv2d_nhwc(x, H, W, wa, 3, 1, 1, colstats=True, frames=frames)
        nm = ops.Normed(y1, ops.ColStats(p1, y1.shape[0], Cout, frames), slope=0.0)
        if res_act:
            y2, p2, _, _ = ops.conv2d_nhwc(nm, H, W, wb, 3, 1, 1, bias=bias, res=res, act=ops.ACT_RELU, colstats=True, frames=frames)
        else:
            y2, p2, _, _ = ops.conv2d_nhwc(nm, H, W, wb, 3, 1, 1, colstats=True, frames=frames)
        return y1.clone(), p1.clone(), y2.clone(), p2.clone()

    try:
        assert hook(force) == 0
        direct = run()
        assert hook(-1) == 0
        implicit = run()
    finally:
        hook(0)
    for d, i in zip(direct, implicit):
        scale = float(i.abs().max())
        assert float((d - i).abs().max()) <= 3e-6 * max(scale, 1.0) * (30 if d.dim() == 3 else 1), (float((d - i).abs().max()), scale)
    ref = torch.nn.functional.conv2d(x.cpu().reshape(frames, H, W, Cin).permute(0, 3, 1, 2).double(), wa.w.cpu().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double(), padding=1)
    refm = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    assert float((direct[0].double().cpu() - refm).abs().max()) < 2e-6 * max(1.0, float(refm.abs().max())) * 4


# ------------------------------------------------------------------------------------------ f16x3: the three-product fp16 split of the large contractions
def _f16_events(ops, reset=True):
    import ctypes

    fn = ops._lib.load().cofi_tune_f16x3_resplit_events
    fn.argtypes, fn.restype = [ctypes.c_int], ctypes.c_long
    return fn(1 if reset else 0)


def _scaled_err(out, ref, scale):
    d = (out.double().cpu() - ref).abs() / scale
    return float(d.max()), float(d.pow(2).mean().sqrt())


# every contraction shape of the KITTI forward that takes the 256 x 128 kernel (rows reduced to a few tiles where the batch-16 row count is
# only repetition), K = 7680 included, plus ragged tiles
F16_GATE_SHAPES = [(2048, 512, 7680), (4096, 1024, 3072), (4096, 256, 3840), (8192, 512, 1536), (8192, 128, 1920), (2048, 2048, 512), (2048, 1024, 2048),
                   (2048, 2048, 1024), (4096, 1024, 256 + 256), (2048, 512, 2048), (2048, 256, 3840), (2048, 512, 1024), (4096 + 77, 256, 1024), (777, 130, 512)]


@pytest.mark.parametrize("M,N,K", F16_GATE_SHAPES)
def test_gemm_f16x3_is_fp32_grade(ops, monkeypatch, M, N, K):
    """THE GATE of the three-product fp16 split (COFI_GEMM_F16X3, csrc/gemm_f16_big.inc): on every shape of the forward that takes the
    256 x 128 kernel, against fp64, its error is at most 2 x the exact-fp32 MFMA kernel's (the bound test_gemm_bf16x6_is_fp32_grade holds
    the six-product split to) - largest scaled deviation and rms -, on operands with a heavy-tailed magnitude distribution as well as on
    normal ones; the pipelined kernel never had to hand a tile to the repair launch; two runs give identical bits."""
    fp, fb = _force_hooks(ops)
    g = torch.Generator().manual_seed(M + 3 * N + K)
    for heavy in (False, True):
        a = torch.randn(M, K, generator=g) * 3
        if heavy:   # magnitudes over ~2^10: the per-panel scale has to place them
            a = a * torch.exp(1.5 * torch.randn(M, K, generator=g))
        w = torch.randn(N, K, generator=g) / K ** 0.5
        bias = torch.randn(N, generator=g)
        ref = a.double() @ w.double().t() + bias.double()
        scale = a.double().abs() @ w.double().abs().t()
        err = {}
        try:
            fb(1, 0)
            for mode, f16, pre in (("f32", False, False), ("bf16x6", False, False), ("bf16x6", True, False), ("bf16x6", True, True)):
                monkeypatch.setattr(ops, "GEMM_MODE", mode)
                monkeypatch.setattr(ops, "F16X3_BIG", f16)
                _f16_events(ops)
                # pre: the weight as a static operand (SplitW) - the kernel reads W's fp16 planes pre-split with panel scales (COFI_GEMM_W_F16PRE)
                wd = ops.presplit(G(w)) if pre else G(w)
                out = ops.gemm(G(a), wd, bias=G(bias))
                key = ("f16x3pre" if pre else "f16x3") if f16 else mode
                err[key] = _scaled_err(out, ref, scale)
                if pre:
                    assert wd._f16pre is not None, "the launch did not take the pre-split weight"
                    assert torch.equal(out, ops.gemm(G(a), wd, bias=G(bias)))
                elif f16:
                    ev = _f16_events(ops)
                    # normal operands never leave the window; the heavy-tailed ones (single entries 2^5 above anything the scouts saw) may
                    # send a few workgroups to the repair launch - their results are held to the same bound below
                    assert ev == 0 or heavy, ("the pipelined kernel handed tiles to the repair launch on ordinary data", ev)
                    assert torch.equal(out, ops.gemm(G(a), G(w), bias=G(bias))), heavy
                    assert not torch.equal(out, x6_out), (heavy, ev)   # ... and it is the three-product kernel that ran
                elif mode == "bf16x6":
                    x6_out = out
        finally:
            fb(0, 0)
        for key in ("f16x3", "f16x3pre"):
            assert err[key][0] < max(2.0 * err["f32"][0], 2e-7) and err[key][1] < 1.5 * err["f32"][1], (key, heavy, err)
            assert err[key][1] < 1.5 * err["bf16x6"][1] + 1e-9, (key, heavy, err)


@pytest.mark.parametrize("M,N,K,frames,ks", [(4096, 256, 512, 1, 1), (4096 + 77, 256, 1024, 1, 1), (8192, 192, 512, 2, 1), (2048, 512, 2560, 2, 2),
                                            (1024, 128, 7680, 1, 3), (333, 130, 96, 1, 1), (20480, 512, 1536, 4, 1),
                                            (512, 128, 32, 1, 1), (768, 256, 64, 2, 1), (256, 128, 128, 1, 1)])
def test_big_tiles_f16x3(ops, monkeypatch, M, N, K, frames, ks):
    """The f16x3 256 x 128 kernel with every epilogue / loader form of the six-product one (test_big_tiles_bf16x6): output, column statistics
    and the normalising loader's result agree with the bf16x6 kernel's to fp32 rounding (both are fp32-grade; the K order is the same, the
    products differ), ragged last tiles, per-frame statistics, split-K; and right against fp64."""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    fp, fb = _force_hooks(ops)
    g = torch.Generator().manual_seed(M + N + K)
    a = G(torch.randn(M, K, generator=g))
    w = G(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = G(torch.randn(N, generator=g))
    N2 = 256
    w2 = ops.presplit(G(torch.randn(N2, N, generator=g) / N ** 0.5))
    gam, bet = G(torch.randn(N, generator=g)), G(torch.randn(N, generator=g))
    groups = 32 if N % 32 == 0 else 2
    per_frame = M % frames == 0 and (M // frames) % 256 == 0
    fr = frames if per_frame else 1

    def run():
        y, part = ops.gemm_colstats(a, w, bias=bias, frames=fr)
        st = ops.ColStats(part, M, groups, fr)
        z = ops.gemm(ops.Normed(y, st, gam, bet, 0.1), w2, frames=fr) if (st.fusable() and N % 32 == 0) else None
        return y.clone(), part.clone(), None if z is None else z.clone()

    try:
        assert fb(1, ks) == 0
        monkeypatch.setattr(ops, "F16X3_BIG", True)
        _f16_events(ops)
        f16 = run()
        assert _f16_events(ops) == 0
        monkeypatch.setattr(ops, "F16X3_BIG", False)
        x6 = run()
    finally:
        fb(0, 0)
    ref = a.double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    scale = a.double().cpu().abs() @ w.double().cpu().abs().t() + bias.double().cpu().abs()
    assert float(((f16[0].double().cpu() - ref).abs() / scale).max()) < 2e-6
    assert not torch.equal(f16[0], x6[0])
    close(f16[1], x6[1], 2e-5 * max(1.0, float(x6[1].abs().max())))
    if f16[2] is not None:
        close(f16[2], x6[2], 2e-4)


def test_f16x3_range_tracking(ops, monkeypatch):
    """The range side of the fp16 split.  The scale of a workgroup's panel comes from its first K-tile, so:
    (a) operands far outside fp16's own range - |x| up to 1e9, or all below 1e-8 - are placed by it: no repair, fp32-grade;
    (b) a K-tile that leaves the window later - an entry 1e4 x larger than anything before (it would overflow to infinity), or tiles 1e5 x
        smaller than the first (their residuals would be flushed) - sends exactly the affected workgroups to the repair launch, whose
        results are fp32-grade again and the same from run to run;
    (c) an all-zero first tile, zero rows;  (d) non-finite inputs come out non-finite in their rows only."""
    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    monkeypatch.setattr(ops, "F16X3_BIG", True)
    fp, fb = _force_hooks(ops)
    g = torch.Generator().manual_seed(77)
    M, N, K = 1024, 256, 1024   # 4 x 2 tiles of 256 x 128
    w = torch.randn(N, K, generator=g) / K ** 0.5

    def check(a, w_, expect_events, tol=2e-6):
        ref = a.double() @ w_.double().t()
        scale = a.double().abs() @ w_.double().abs().t() + 1e-300
        _f16_events(ops)
        out = ops.gemm(G(a), G(w_))
        ev = _f16_events(ops)
        assert (ev > 0) == expect_events if isinstance(expect_events, bool) else ev == expect_events, ev
        fin = torch.isfinite(ref)
        e = ((out.double().cpu() - ref).abs() / scale)[fin]
        assert float(e.max()) < tol, float(e.max())
        assert torch.equal(out, ops.gemm(G(a), G(w_)))
        return out, ev

    try:
        fb(1, 1)
        a0 = torch.randn(M, K, generator=g)
        check(a0, w, False)
        # (a) magnitudes no fp16 could hold, placed by the first tile
        check(a0 * 1e9, w, False)
        check(a0 * 1e-9, w * 1e-6, False)
        # (b) one entry 1e4 x the rest in K-tile 20 of row panel 1: its two workgroups (two column tiles) are repaired, the others are not
        a1 = a0.clone()
        a1[300, 20 * 32 + 5] = 3e4
        check(a1, w, 2)
        # ... tiles far BELOW the window: columns 512.. of row panel 2 are 1e-6 x the first tiles
        a2 = a0.clone()
        a2[512:768, 512:] *= 1e-6
        check(a2, w, 2)
        # ... the same on the W side: one column panel
        w1 = w.clone()
        w1[130, 700] = 500.0
        check(a0, w1, 4)
        # a panel whose rows differ by 2^20 in magnitude: the small rows keep an fp32-grade RELATIVE error down to 2^-12 of the panel maximum
        a3 = a0.clone()
        a3[0:64] *= 2.0 ** -12
        check(a3, w, False, tol=4e-6)
        # (c) zeros
        a4 = a0.clone()
        a4[:, :32] = 0
        a4[100:110] = 0
        check(a4, w, False)
        # (d) non-finite inputs stay in their rows
        a5 = a0.clone()
        a5[5, 40] = float("inf")
        a5[700, 900] = float("nan")
        out = ops.gemm(G(a5), G(w)).cpu()
        bad = ~torch.isfinite(out).all(1)
        assert bad[5] and bad[700] and int(bad.sum()) == 2
        good = ~bad
        ref = a5.double() @ w.double().t()
        scale = a5.double().abs() @ w.double().abs().t()
        assert float(((out.double() - ref).abs() / scale)[good].max()) < 2e-6
    finally:
        fb(0, 0)


@pytest.mark.parametrize("pad,cmid", [(1, 128), (0, 128)])
def test_big_tiles_convolution_f16x3(ops, monkeypatch, pad, cmid):
    """3 x 3 / stride 1 convolutions on the f16x3 256 x 128 kernel, pending InstanceNorm + ReLU of the producer applied by the loader
    included, stack mode: agrees with the six-product kernel to fp32 rounding and with torch's convolution."""
    from cofii2p_amd.image import _nhwc_weight

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    fp, fb = _force_hooks(ops)
    frames, Cin, Cmid, Cout, H, W = 2, 64, cmid, 256, 40 + 2 * (1 - pad), 128 + 2 * (1 - pad)
    g = torch.Generator().manual_seed(12)
    x = G(torch.randn(frames * H * W, Cin, generator=g))
    wa = ops.presplit(G(_nhwc_weight(torch.randn(Cmid, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)))
    wb = ops.presplit(G(_nhwc_weight(torch.randn(Cout, Cmid, 3, 3, generator=g) / (Cmid * 9) ** 0.5)))

    def run():
        y1, p1, Ho, Wo = ops.conv2d_nhwc(x, H, W, wa, 3, 1, pad, colstats=True, frames=frames)
        if pad == 1:
            nm = ops.Normed(y1, ops.ColStats(p1, y1.shape[0], Cmid, frames), slope=0.0)   # InstanceNorm + ReLU pending
            y2, p2, _, _ = ops.conv2d_nhwc(nm, Ho, Wo, wb, 3, 1, 1, colstats=True, frames=frames)
            return y1.clone(), p1.clone(), y2.clone(), p2.clone()
        return y1.clone(), p1.clone()

    try:
        assert fb(1, 1) == 0
        monkeypatch.setattr(ops, "F16X3_BIG", True)
        _f16_events(ops)
        r16 = run()
        assert _f16_events(ops) == 0
        monkeypatch.setattr(ops, "F16X3_BIG", False)
        r6 = run()
    finally:
        fb(0, 0)
    assert not torch.equal(r16[0], r6[0])
    close(r16[0], r6[0], 2e-5)
    close(r16[1], r6[1], 2e-5 * max(1.0, float(r6[1].abs().max())))
    if pad == 1:
        close(r16[2], r6[2], 3e-4)
    y16 = r16[0]
    ref = torch.nn.functional.conv2d(x.cpu().reshape(frames, H, W, Cin).permute(0, 3, 1, 2), wa.w.cpu().reshape(Cmid, 3, 3, Cin).permute(0, 3, 1, 2), padding=pad)
    close(y16, ref.permute(0, 2, 3, 1).reshape(-1, Cmid), 2e-5)


def test_f16x3_presplit_weight_is_refused_where_the_kernel_does_not_run(ops, monkeypatch):
    """COFI_GEMM_W_F16PRE is readable by gemm_f16_big_kernel only: cofi_gemm_f16x3_eligible says which launches run on it, ops passes the
    pre-split form exactly there, and the C ABI REFUSES it (COFI_EUNSUPPORTED, nothing launched) on any other launch instead of
    multiplying fp16 pairs as if they were fp32"""
    import ctypes

    monkeypatch.setattr(ops, "GEMM_MODE", "bf16x6")
    monkeypatch.setattr(ops, "F16X3_BIG", True)
    lib = ops._lib.load()
    g = torch.Generator().manual_seed(2)
    for (M, N, K), want in (((40960, 256, 1024), 1), ((512, 64, 256), 0), ((40960, 256, 100), 0), ((1280, 128, 128), 0)):
        assert lib.cofi_gemm_f16x3_eligible(M, N, K, 0, 1) == want, (M, N, K)
        a = G(torch.randn(M, K, generator=g))
        w = ops.presplit(G(torch.randn(N, K, generator=g) / K ** 0.5))
        out = ops.gemm(a, w)
        assert (w._f16pre is not None) == bool(want), (M, N, K)            # ops hands over the pre-split form only where it is eligible
        close(out, (a.cpu().double() @ w.w.cpu().double().t()).float(), 2e-5 * K ** 0.5)
        if not want and K % 4 == 0:
            c = torch.empty((M, N), device=DEV)
            ws = ops._WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, N, K), a.device)
            flags = ops.GEMM_BF16X6 | ops.GEMM_F16X3 | ops.GEMM_W_F16PRE
            rc = lib.cofi_gemm_f32_fused(ops._p(a), K, None, ops._p(w.f16pre), K, ops._p(c), N, M, N, K, None, None, flags, None, 1, ops._p(ws),
                                         0 if ws is None else ws.numel(), 1, ops._stream())
            assert rc == -3, rc   # COFI_EUNSUPPORTED
