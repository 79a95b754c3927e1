"""Row f1 on the GPU: cofi_pnp_ransac against the oracle (same samples -> same hypotheses) and against ground truth."""
import numpy as np
import pytest
import torch

import pnp_oracle as po
from test_pose_cpu import K, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pose_mod():
    from cofii2p_amd import pose
    return pose


@pytest.mark.parametrize("seed,noise,outl", [(0, 0.5, 0.3), (1, 1.0, 0.5), (2, 0.0, 0.0)])
def test_pnp_matches_oracle_and_ground_truth(pose_mod, seed, noise, outl):
    rng = np.random.default_rng(10 + seed)
    X, uv, P, inl = synth(rng, n=500, noise=noise, outliers=outl)
    iters = 512
    res, R, t, mask = pose_mod.solve_pnp_ransac(torch.from_numpy(X).to(DEV), torch.from_numpy(uv).to(DEV), K, iterations=iters, seed=5)
    res = res.cpu().numpy()
    assert res[0] == 1
    ok, Ro, to, masko, hyp = po.solve_pnp_ransac(X, uv, K, iterations=iters, seed=5)
    assert ok
    # same samples, same solver: the winning consensus set has the same size up to points that sit on the 8 px boundary
    assert abs(int(res[1]) - int(masko.sum())) <= 3
    Pg, Po = pose_mod.pose_matrix(R, t), np.eye(4)
    Po[:3, :3], Po[:3, 3] = Ro, to
    d_t, d_r = pose_mod.get_P_diff(Pg, Po)
    assert d_t < 2e-3 and d_r < 2e-2, (d_t, d_r)                 # GPU vs oracle (fp32 scoring vs fp64)
    rte, rre = pose_mod.get_P_diff(Pg, P)
    assert rte < (1e-4 if noise == 0 else 0.05) and rre < (1e-3 if noise == 0 else 0.2), (rte, rre)   # vs ground truth
    m = mask.cpu().numpy().astype(bool)
    assert m[inl].mean() > (0.97 if noise == 0 else 0.6) and (outl == 0 or m[~inl].mean() < 0.1)


def test_pnp_device_count_and_failure(pose_mod):
    rng = np.random.default_rng(3)
    X, uv, P, _ = synth(rng, n=300, noise=0.3, outliers=0.2)
    Xg, ug = torch.from_numpy(X).to(DEV), torch.from_numpy(uv).to(DEV)
    cnt = torch.tensor([200], dtype=torch.int32, device=DEV)
    res, R, t, mask = pose_mod.solve_pnp_ransac(Xg, ug, K, iterations=256, seed=1, count=cnt)
    res2, R2, t2, mask2 = pose_mod.solve_pnp_ransac(Xg[:200].contiguous(), ug[:200].contiguous(), K, iterations=256, seed=1)
    assert torch.equal(res.cpu(), res2.cpu()) and torch.equal(R, R2) and torch.equal(t, t2)   # bit-reproducible
    assert int(mask[200:].sum()) == 0
    cnt3 = torch.tensor([3], dtype=torch.int32, device=DEV)
    res3, *_ = pose_mod.solve_pnp_ransac(Xg, ug, K, iterations=64, count=cnt3)
    assert int(res3[0]) == 0


def test_pose_of_a_forward_frame(pose_mod):
    """end to end: the fine matches of a synthetic frame give SOME pose; the call is sync-free and capacity-sized"""
    from cofii2p_amd.network import CoFiI2P
    import bench

    model = CoFiI2P(bench.Opt()).to(DEV)
    pyr, img, fr = bench.make_inputs(torch.device(DEV), [0], 20480)[0]
    out = model(pyr, img, None, None, None, "test")
    fine_xy = model.last_match["fine_xy"]          # (2, n)
    n = out[7].shape[0]
    assert fine_xy.shape == (2, n)
    Kc = np.array([[300.0, 0, 256.0], [0, 300.0, 80.0], [0, 0, 1.0]])
    res, R, t, mask = pose_mod.solve_pnp_ransac(out[7].contiguous(), fine_xy.t().contiguous(), Kc, iterations=2000)
    res = res.cpu().numpy()
    assert res[0] in (0, 1) and (res[0] == 0 or res[1] >= 4)
    Rm = R.cpu().numpy().astype(np.float64)
    if res[0]:
        assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-4) and np.linalg.det(Rm) > 0.99
