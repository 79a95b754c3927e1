"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the pose step of the reference's evaluation,
evaluation/eval_all.py:107-117:

    is_success, R, t, inliers = cv2.solvePnPRansac(cameraMatrix=K, imagePoints=fine_xy.T, objectPoints=coarse_pc_points,
                                                   iterationsCount=10000, distCoeffs=None)   # reprojectionError 8, ITERATIVE refit
    t_diff, angles_diff = get_P_diff(T_pred, P)                                              # eval_all.py:16-22

PARITY UNPINNED: cv2 (OpenCV) is a third-party dependency that is absent from this image and from /root/reference, so
neither its RANSAC sampling nor its solver can be run here.  This file restates the published algorithm class OpenCV
documents for that call — minimal-set hypotheses scored by reprojection error (threshold 8 px), best consensus set,
Levenberg-Marquardt refit of the reprojection error on its inliers — with a P3P (Grunert) minimal solver on 3 points and
the 4th sampled point choosing among its up-to-4 solutions.  It is validated geometrically (synthetic poses, noise,
outliers: tests/test_pose_cpu.py) and `get_P_diff` is pinned against scipy's Rotation exactly as the reference computes it.
The HIP implementation (cofii2p_amd/csrc/pnp.hip) draws the SAME samples (counter-based hash below), so GPU and oracle
score the same hypotheses.
"""
import numpy as np

MASK32 = 0xFFFFFFFF


def hash_u32(seed: int, hyp: int, j: int) -> int:
    """Counter-based generator shared with the HIP kernel (lowbias32 of a mixed counter)."""
    x = (seed * 0x9E3779B1 + hyp * 0x85EBCA77 + j * 0xC2B2AE3D + 0x27D4EB2F) & MASK32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & MASK32
    x ^= x >> 15
    x = (x * 0x846CA68B) & MASK32
    x ^= x >> 16
    return x


def sample4(seed: int, hyp: int, n: int):
    """4 distinct indices in [0, n): draw j-th candidate, on a repeat take the next unused index cyclically."""
    idx = []
    for j in range(4):
        c = hash_u32(seed, hyp, j) % n
        while c in idx:
            c = (c + 1) % n
        idx.append(c)
    return idx


def _poly_mul(a, b):
    return np.convolve(a, b)


def p3p_grunert(P, f):
    """P (3,3) world points, f (3,3) unit bearing vectors -> list of (R, t) with R P_i + t = s_i f_i, s_i > 0.
    Depth ratios u = s2/s1, v = s3/s1: eliminating u from the two conics gives a quartic in v (coefficients by polynomial
    arithmetic; Haralick et al. 1994, Grunert's solution)."""
    a2 = np.sum((P[1] - P[2]) ** 2)
    b2 = np.sum((P[0] - P[2]) ** 2)
    c2 = np.sum((P[0] - P[1]) ** 2)
    if min(a2, b2, c2) < 1e-18:
        return []
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    # polynomials in v, lowest degree first
    w = np.array([1.0, -2.0 * cb, 1.0])                    # 1 - 2 v cos(beta) + v^2
    N = (a2 - c2) * w - b2 * np.array([-1.0, 0.0, 1.0])    # numerator of u
    D = 2.0 * b2 * np.array([cg, -ca])                     # denominator of u
    D2, N2, ND = _poly_mul(D, D), _poly_mul(N, N), _poly_mul(N, D)
    q = b2 * (np.pad(D2, (0, 2)) + N2 - 2.0 * cg * np.pad(ND, (0, 1))) - c2 * _poly_mul(w, D2)
    roots = np.roots(q[::-1])
    sols = []
    for r in roots:
        if abs(r.imag) > 1e-6 * max(1.0, abs(r.real)):
            continue
        v = r.real
        for _ in range(3):  # Newton polish on the real quartic
            pv = np.polyval(q[::-1], v)
            dv = np.polyval(np.polyder(q[::-1]), v)
            if dv != 0:
                v -= pv / dv
        den = D[0] + D[1] * v
        if abs(den) < 1e-12 or v <= 0:
            continue
        u = (N[0] + N[1] * v + N[2] * v * v) / den
        wv = 1.0 - 2.0 * v * cb + v * v
        if u <= 0 or wv <= 0:
            continue
        s1 = np.sqrt(b2 / wv)
        C = np.stack([s1 * f[0], u * s1 * f[1], v * s1 * f[2]])
        Rt = _triad(P, C)
        if Rt is not None:
            sols.append(Rt)
    return sols


def _frame(A):
    e1 = A[1] - A[0]
    n1 = np.linalg.norm(e1)
    e3 = np.cross(e1, A[2] - A[0])
    n3 = np.linalg.norm(e3)
    if n1 < 1e-12 or n3 < 1e-12:
        return None
    e1, e3 = e1 / n1, e3 / n3
    return np.stack([e1, np.cross(e3, e1), e3], 1)


def _triad(P, C):
    Fp, Fc = _frame(P), _frame(C)
    if Fp is None or Fc is None:
        return None
    R = Fc @ Fp.T
    return R, C[0] - R @ P[0]


def project(R, t, X, K4):
    fx, fy, cx, cy = K4
    Y = X @ R.T + t
    z = Y[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.stack([fx * Y[:, 0] / z + cx, fy * Y[:, 1] / z + cy], 1)
    return u, z


def reproj_err2(R, t, X, uv, K4):
    u, z = project(R, t, X, K4)
    e = np.sum((u - uv) ** 2, 1)
    e[~(z > 1e-6)] = np.inf
    return e


def hypothesis(seed, hyp, X, uv, K4):
    """The pose of hypothesis `hyp`: P3P on samples 0..2, sample 3 picks among the solutions (smallest reprojection error)."""
    n = X.shape[0]
    i = sample4(seed, hyp, n)
    fx, fy, cx, cy = K4
    b = np.stack([(uv[i[:3], 0] - cx) / fx, (uv[i[:3], 1] - cy) / fy, np.ones(3)], 1)
    f = b / np.linalg.norm(b, axis=1, keepdims=True)
    best = None
    for R, t in p3p_grunert(X[i[:3]].astype(np.float64), f):
        e = reproj_err2(R, t, X[i[3:4]].astype(np.float64), uv[i[3:4]].astype(np.float64), K4)[0]
        if np.isfinite(e) and (best is None or e < best[0]):
            best = (e, R, t)
    return None if best is None else (best[1], best[2])


def refine(R, t, X, uv, K4, iters=20):
    """Levenberg-Marquardt on the reprojection error over the given points, left-multiplied se(3) increments."""
    fx, fy, cx, cy = K4
    lam = 1e-3

    def cost_and_normal(R, t):
        Y = X @ R.T + t
        x, y, z = Y[:, 0], Y[:, 1], Y[:, 2]
        r = np.stack([fx * x / z + cx - uv[:, 0], fy * y / z + cy - uv[:, 1]], 1)
        J = np.zeros((X.shape[0], 2, 6))
        # d(pi)/dY
        a = np.zeros((X.shape[0], 2, 3))
        a[:, 0, 0], a[:, 0, 2] = fx / z, -fx * x / z ** 2
        a[:, 1, 1], a[:, 1, 2] = fy / z, -fy * y / z ** 2
        # dY = omega x Y + delta  ->  [-[Y]x | I]
        S = np.zeros((X.shape[0], 3, 3))
        S[:, 0, 1], S[:, 0, 2] = z, -y
        S[:, 1, 0], S[:, 1, 2] = -z, x
        S[:, 2, 0], S[:, 2, 1] = y, -x
        J[:, :, :3] = a @ S
        J[:, :, 3:] = a
        Jf, rf = J.reshape(-1, 6), r.reshape(-1)
        return float(rf @ rf), Jf.T @ Jf, Jf.T @ rf

    c, H, g = cost_and_normal(R, t)
    for _ in range(iters):
        d = np.linalg.solve(H + lam * np.diag(np.diag(H)) + 1e-12 * np.eye(6), -g)
        Rn, tn = se3_update(R, t, d)
        cn, Hn, gn = cost_and_normal(Rn, tn)
        if np.isfinite(cn) and cn < c:
            R, t, c, H, g, lam = Rn, tn, cn, Hn, gn, max(lam * 0.1, 1e-9)
        else:
            lam = min(lam * 10.0, 1e6)
    return R, t


def se3_update(R, t, d):
    w, dt = d[:3], d[3:]
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        E = np.eye(3) + Wx
    else:
        E = np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * (Wx @ Wx)
    return E @ R, E @ t + dt


def solve_pnp_ransac(X, uv, K, iterations=10000, reproj_error=8.0, seed=0, refine_iters=20):
    """-> (success, R (3,3), t (3,), inlier mask (n,), best hypothesis id).  X (n,3), uv (n,2), K (3,3)."""
    X, uv = np.asarray(X, np.float64), np.asarray(uv, np.float64)
    n = X.shape[0]
    K4 = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
    if n < 4:
        return False, np.eye(3), np.zeros(3), np.zeros(n, bool), -1
    best = (-1, -1, None)
    thr2 = reproj_error ** 2
    for h in range(iterations):
        Rt = hypothesis(seed, h, X, uv, K4)
        if Rt is None:
            continue
        cnt = int(np.sum(reproj_err2(Rt[0], Rt[1], X, uv, K4) <= thr2))
        if cnt > best[0]:
            best = (cnt, h, Rt)
    if best[2] is None or best[0] < 4:
        return False, np.eye(3), np.zeros(3), np.zeros(n, bool), -1
    R, t = best[2]
    mask = reproj_err2(R, t, X, uv, K4) <= thr2
    R, t = refine(R, t, X[mask], uv[mask], K4, refine_iters)
    return True, R, t, mask, best[1]


def euler_xzy_deg(Rm):
    """scipy Rotation.from_matrix(Rm).as_euler('xzy', degrees=True) (extrinsic x, then z, then y): R = Ry(c) Rz(b) Rx(a)."""
    b = np.arcsin(np.clip(Rm[1, 0], -1.0, 1.0))
    if abs(Rm[1, 0]) < 1 - 1e-12:
        a = np.arctan2(-Rm[1, 2], Rm[1, 1])
        c = np.arctan2(-Rm[2, 0], Rm[0, 0])
    else:  # gimbal lock: scipy sets the third angle to zero
        a = np.arctan2(Rm[2, 1], Rm[2, 2])
        c = 0.0
    return np.degrees(np.array([a, b, c]))


def get_P_diff(P_pred, P_gt):
    """eval_all.py:16-22: (RTE, RRE) = (|t|, sum |euler xzy, degrees|) of inv(P_pred) @ P_gt."""
    P_diff = np.linalg.inv(P_pred) @ P_gt
    return float(np.linalg.norm(P_diff[:3, 3])), float(np.sum(np.abs(euler_xzy_deg(P_diff[:3, :3]))))
