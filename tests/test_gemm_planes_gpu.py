"""gemm_planes_kernel (csrc/gemm_planes.inc: both operands as bf16 hi / lo planes, LDS-DMA pipeline) on the GPU:
every configuration x split-K against the register-staged bf16x3 kernel at the same split - output bit for bit, statistics partials to
rounding (the row-pass order of the epilogue depends on the workgroup shape) - and against an fp64 product.  Shapes cover row / column
tails, K tails (K % K-tile != 0, K < one tile), N = 1 and M smaller than a tile.
Needs a real MI355X:  python -m pytest tests -m gpu"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NUM_CFG = 9   # kPlanesCfg of csrc/gemm.hip


@pytest.fixture(scope="module")
def hooks():
    from cofii2p_amd import _lib, ops

    lib = _lib.load()
    fp, fo = lib.cofi_tune_force_planes, lib.cofi_tune_force_plan   # tuning hooks (not part of the public header)
    fp.argtypes, fp.restype, fo.argtypes, fo.restype = [ctypes.c_int] * 2, ctypes.c_int, [ctypes.c_int] * 3, ctypes.c_int
    saved, ops.GEMM_MODE = ops.GEMM_MODE, "bf16x3"
    yield ops, fp, fo
    fp(-1, 0), fo(0, 0, 0)
    ops.GEMM_MODE = saved


def split_a(ops, a):
    w = ops.SplitW(a)   # the library's own split (cofi_split_bf16_planes): planes (2, M, ld)
    return ops.SplitA(w.planes, a.shape[1])


@pytest.mark.parametrize("M,N,K", [(200, 96, 200), (64, 32, 480), (1280, 512, 1024), (333, 130, 72), (2560, 256, 3840), (4096, 64, 960), (130, 1, 64),
                                   (777, 513, 1928), (20480, 32, 480)])
def test_every_configuration_equals_the_register_staged_kernel(hooks, M, N, K):
    ops, fp, fo = hooks
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + 7 * K)
    a = torch.randn((M, K), device=DEV, generator=g)
    w = torch.randn((N, K), device=DEV, generator=g) / K ** 0.5
    bias = torch.randn((N,), device=DEV, generator=g)
    rowdiv = torch.randint(1, 9, (M,), device=DEV, generator=g).float()
    sa, sw = split_a(ops, a), ops.SplitW(w)
    ref64 = a.double() @ w.double().t()
    width = 32 if N % 32 == 0 else (2 if N % 2 == 0 else 1)
    try:
        for ks in (1, 3):
            if ks > 1 and K < 512:
                continue
            fp(-2, 0), fo(64, 64, ks)   # pre-split operands through the register-staged kernel, 64 x 64 tiles
            ref, refp = ops.gemm_colstats(sa, sw, bias=bias, rowdiv=rowdiv, act=ops.ACT_LEAKY01, stat_width=width)
            raw = ops.gemm(sa, sw)
            assert float((raw - ref64).abs().max() / ref64.abs().max()) < 2e-5   # 3-term bf16 split: ~2^-16 per product
            fo(0, 0, 0)
            for cfg in range(NUM_CFG):
                fp(cfg, ks)
                y, part = ops.gemm_colstats(sa, sw, bias=bias, rowdiv=rowdiv, act=ops.ACT_LEAKY01, stat_width=width)
                assert torch.equal(y, ref), (cfg, ks)
                assert torch.allclose(part, refp, rtol=2e-5, atol=1e-3), (cfg, ks)
                assert torch.equal(ops.gemm(sa, sw), raw), (cfg, ks)
        fp(-1, 0)   # default plan (table + heuristic), whatever split it takes
        y = ops.gemm(sa, sw)
        assert float((y - ref64).abs().max() / ref64.abs().max()) < 2e-5
    finally:
        fp(-1, 0), fo(0, 0, 0)


def test_planes_gemm_is_deterministic_and_leaves_neighbours_alone(hooks):
    """run-to-run identical bits (no atomics, fixed-order split-K reduction), output written into a column slice of a wider buffer"""
    ops, fp, fo = hooks
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn((1280, 3840), device=DEV, generator=g)
    w = torch.randn((256, 3840), device=DEV, generator=g) / 62.0
    sa, sw = split_a(ops, a), ops.SplitW(w)
    buf = torch.full((1280, 512), 7.0, device=DEV)
    y0 = ops.gemm(sa, sw, out=buf[:, 128:384]).clone()
    for _ in range(3):
        assert torch.equal(ops.gemm(sa, sw, out=buf[:, 128:384]), y0)
    assert float((buf[:, :128] - 7.0).abs().max()) == 0.0 and float((buf[:, 384:] - 7.0).abs().max()) == 0.0
