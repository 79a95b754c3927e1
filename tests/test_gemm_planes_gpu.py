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


def test_splitk_gemms_are_bit_reproducible_under_concurrent_load(hooks):
    """Split-K GEMMs (slabs + the fixed-order reduction launch) of both kernels, several tile shapes, 150 launches on four streams at
    once - every stream with its own workspace: each output must equal, bit for bit, the one computed alone.  (Round 3 also tried the
    reduction INSIDE the launch - last-arriving workgroup, agent-scope release / acquire; it passed this test and lost 17 % of the
    frame rate: the per-workgroup L2 write-back of the release hits the other frames' kernels.  DESIGN.md section 6.)"""
    ops, fp, fo = hooks
    g = torch.Generator(device=DEV).manual_seed(11)
    probs = []
    for (M, N, K, cfg, ks) in ((1280, 512, 7680, 0, 6), (1280, 256, 3840, 1, 6), (2560, 256, 3840, 1, 3), (320, 256, 2304, 1, 6), (2560, 128, 1920, 6, 3)):
        a = torch.randn((M, K), device=DEV, generator=g)
        w = torch.randn((N, K), device=DEV, generator=g) / K ** 0.5
        probs.append((a, split_a(ops, a), ops.SplitW(w), torch.randn((N,), device=DEV, generator=g), cfg, ks))
    streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]

    def run(pr, planes):
        a, sa, sw, bias, cfg, ks = pr
        if planes:
            fp(cfg, ks)
            return ops.gemm_colstats(sa, sw, bias=bias, act=ops.ACT_LEAKY01, stat_width=32)
        fo(64, 64, ks)
        return ops.gemm_colstats(a, sw, bias=bias, act=ops.ACT_LEAKY01, stat_width=32)

    try:
        want = {}
        for i, pr in enumerate(probs):
            for planes in (False, True):
                want[(i, planes)] = [t.clone() for t in run(pr, planes)]
        torch.cuda.synchronize()
        outs = []
        for rep in range(150):
            i, planes = rep % len(probs), bool((rep // len(probs)) & 1)
            with torch.cuda.stream(streams[rep % 4]):
                outs.append(((i, planes), run(probs[i], planes)))
        torch.cuda.synchronize()
        for key, (y, part) in outs:
            assert torch.equal(y, want[key][0]) and torch.equal(part, want[key][1]), key
    finally:
        fp(-1, 0), fo(0, 0, 0)
