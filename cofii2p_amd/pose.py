"""Camera pose from the fine matches + the registration errors the reference's evaluation reports
(evaluation/eval_all.py:16-22, 107-117) — row f1 of SURVEY.md §8.

    ok, R, t, inliers = solve_pnp_ransac(coarse_pc_points, fine_xy.T, K)          # cv2.solvePnPRansac(..., iterationsCount=10000)
    rte, rre = get_P_diff(T_pred, P_gt)                                            # eval_all.py:16-22

`solve_pnp_ransac` runs on the device (cofi_pnp_ransac: P3P hypotheses in parallel, consensus by reprojection error, LM refit);
its arguments are CUDA tensors and nothing is copied to the host unless the caller asks for `ok`.  OpenCV is not available in
this environment, so equality with cv2's result is not pinned; the implementation is checked against
oracle/pnp_oracle.py and against ground-truth poses of synthetic correspondences."""
from typing import Optional

import numpy as np
import torch

from . import _lib, ops


def solve_pnp_ransac(object_points: torch.Tensor, image_points: torch.Tensor, K, iterations: int = 10000, reproj_error: float = 8.0,
                     seed: int = 0, refine_iters: int = 20, count: Optional[torch.Tensor] = None):
    """object_points (n,3), image_points (n,2) float32 CUDA, K (3,3) (tensor / array: fx, fy, cx, cy are read on the host).
    count (optional int32 device tensor): number of valid rows, read on the device (capacity-sized inputs of the test-mode
    forward).  Returns (result, R (3,3), t (3,), inlier_mask (n,) uint8) as device tensors; result = int32 [success, inliers,
    winning hypothesis]."""
    lib = _lib.load()
    for t_, name, w in ((object_points, "object_points", 3), (image_points, "image_points", 2)):
        if not t_.is_cuda or t_.dtype != torch.float32 or t_.dim() != 2 or t_.shape[1] != w or not t_.is_contiguous():
            raise _lib.CofiError("solve_pnp_ransac: %s must be a contiguous CUDA float32 (n,%d) tensor" % (name, w))
    n = object_points.shape[0]
    if image_points.shape[0] != n or n == 0:
        raise _lib.CofiError("solve_pnp_ransac: need n >= 1 correspondences of equal count")
    Kh = K.detach().cpu().numpy() if torch.is_tensor(K) else np.asarray(K)
    dev = object_points.device
    ws = torch.empty(lib.cofi_pnp_ransac_workspace(iterations), dtype=torch.uint8, device=dev)
    pose = torch.empty(12, dtype=torch.float32, device=dev)
    result = torch.empty(3, dtype=torch.int32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    rc = lib.cofi_pnp_ransac(ops._p(object_points), ops._p(image_points), ops._p(count), n, float(Kh[0, 0]), float(Kh[1, 1]),
                             float(Kh[0, 2]), float(Kh[1, 2]), int(iterations), float(reproj_error), int(seed) & 0xFFFFFFFF,
                             int(refine_iters), ops._p(ws), ws.numel(), ops._p(pose), ops._p(result), ops._p(mask), ops._stream())
    _lib.check(rc, "cofi_pnp_ransac")
    return result, pose[:9].view(3, 3), pose[9:], mask


def euler_xzy_deg(Rm: np.ndarray) -> np.ndarray:
    """scipy's Rotation.from_matrix(Rm).as_euler('xzy', degrees=True) (eval_all.py:20-21) without scipy: R = Ry(c) Rz(b) Rx(a)."""
    b = np.arcsin(np.clip(Rm[1, 0], -1.0, 1.0))
    if abs(Rm[1, 0]) < 1 - 1e-12:
        a = np.arctan2(-Rm[1, 2], Rm[1, 1])
        c = np.arctan2(-Rm[2, 0], Rm[0, 0])
    else:  # gimbal lock: third angle set to zero, as scipy does
        a = np.arctan2(Rm[2, 1], Rm[2, 2])
        c = 0.0
    return np.degrees(np.array([a, b, c]))


def get_P_diff(P_pred_np: np.ndarray, P_gt_np: np.ndarray):
    """eval_all.py:16-22: (t_diff, angles_diff) = (RTE, RRE) of inv(P_pred) @ P_gt."""
    P_diff = np.dot(np.linalg.inv(P_pred_np), P_gt_np)
    t_diff = np.linalg.norm(P_diff[0:3, 3])
    angles_diff = np.sum(np.abs(euler_xzy_deg(P_diff[0:3, 0:3])))
    return t_diff, angles_diff


def pose_matrix(R: torch.Tensor, t: torch.Tensor) -> np.ndarray:
    """T_pred of eval_all.py:111-113."""
    T = np.eye(4)
    T[0:3, 0:3] = R.detach().cpu().numpy().astype(np.float64)
    T[0:3, 3] = t.detach().cpu().numpy().astype(np.float64)
    return T
