"""Row f1: the pose oracle (oracle/pnp_oracle.py) validated geometrically, and get_P_diff pinned exactly as the reference computes
it (evaluation/eval_all.py:16-22 uses scipy's Rotation).  cv2 is absent: parity with cv2.solvePnPRansac itself is unpinned."""
import numpy as np
from scipy.spatial.transform import Rotation

import pnp_oracle as po
from cofii2p_amd import pose

K = np.array([[700.0, 0, 256.0], [0, 700.0, 80.0], [0, 0, 1.0]])


def synth(rng, n=400, noise=1.0, outliers=0.4):
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.4).as_matrix()
    t = rng.normal(size=3) * 2 + np.array([0, 0, 12.0])
    X = rng.uniform(-20, 20, (n, 3))
    X[:, 2] = rng.uniform(-5, 5, n)
    Y = X @ R.T + t
    uv = np.stack([K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]], 1) + rng.normal(size=(n, 2)) * noise
    out = rng.random(n) < outliers
    uv[out] = rng.uniform(0, 512, (int(out.sum()), 2))
    P = np.eye(4)
    P[:3, :3], P[:3, 3] = R, t
    return X.astype(np.float32), uv.astype(np.float32), P, ~out


def reference_get_P_diff(P_pred_np, P_gt_np):
    """the reference's six lines, verbatim in meaning (scipy available here)"""
    P_diff = np.dot(np.linalg.inv(P_pred_np), P_gt_np)
    t_diff = np.linalg.norm(P_diff[0:3, 3])
    angles_diff = np.sum(np.abs(Rotation.from_matrix(P_diff[0:3, 0:3]).as_euler("xzy", degrees=True)))
    return t_diff, angles_diff


def test_get_P_diff_equals_reference_formula():
    rng = np.random.default_rng(0)
    for _ in range(200):
        A, B = np.eye(4), np.eye(4)
        A[:3, :3] = Rotation.random(random_state=int(rng.integers(1 << 31))).as_matrix()
        B[:3, :3] = Rotation.random(random_state=int(rng.integers(1 << 31))).as_matrix()
        A[:3, 3], B[:3, 3] = rng.normal(size=3), rng.normal(size=3)
        ref = reference_get_P_diff(A, B)
        for fn in (pose.get_P_diff, po.get_P_diff):
            got = fn(A, B)
            assert abs(got[0] - ref[0]) < 1e-12 and abs(got[1] - ref[1]) < 1e-9
    # gimbal lock (middle angle +-90 degrees): same convention as scipy (third angle zero)
    import warnings

    G = np.eye(4)
    G[:3, :3] = Rotation.from_euler("xzy", [20, 90, 0], degrees=True).as_matrix()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # scipy announces the gimbal lock
        ref = reference_get_P_diff(np.eye(4), G)[1]
    assert abs(pose.get_P_diff(np.eye(4), G)[1] - ref) < 1e-6


def test_p3p_recovers_exact_poses():
    rng = np.random.default_rng(1)
    hits = 0
    for _ in range(100):
        R = Rotation.from_rotvec(rng.normal(size=3) * 0.7).as_matrix()
        t = rng.normal(size=3) * 2 + np.array([0, 0, 10.0])
        X = rng.uniform(-5, 5, (3, 3))
        Y = X @ R.T + t
        f = Y / np.linalg.norm(Y, axis=1, keepdims=True)
        sols = po.p3p_grunert(X, f)
        assert 1 <= len(sols) <= 4
        hits += min(np.abs(Rs - R).max() + np.abs(ts - t).max() for Rs, ts in sols) < 1e-6
    assert hits == 100


def test_sampling_is_distinct_and_reproducible():
    for n in (4, 5, 17, 400):
        for h in range(50):
            i = po.sample4(7, h, n)
            assert len(set(i)) == 4 and all(0 <= k < n for k in i) and i == po.sample4(7, h, n)


def test_ransac_recovers_pose_with_outliers():
    rng = np.random.default_rng(2)
    for noise, outl in ((0.5, 0.3), (1.0, 0.5)):
        X, uv, P, inl = synth(rng, noise=noise, outliers=outl)
        ok, R, t, mask, _ = po.solve_pnp_ransac(X, uv, K, iterations=400, seed=3)
        assert ok
        Pe = np.eye(4)
        Pe[:3, :3], Pe[:3, 3] = R, t
        rte, rre = po.get_P_diff(Pe, P)
        assert rte < 0.05 and rre < 0.2, (rte, rre)
        # the mask is the consensus set of the (unrefined) minimal-sample winner, as cv2 returns it: most true inliers, few outliers
        assert mask[inl].mean() > 0.6 and mask[~inl].mean() < 0.1


def test_degenerate_inputs():
    ok, *_ = po.solve_pnp_ransac(np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32), K, iterations=10)
    assert not ok
    X = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (20, 1))  # all points coincide: no hypothesis exists
    ok, *_ = po.solve_pnp_ransac(X, np.zeros((20, 2), np.float32), K, iterations=20)
    assert not ok
