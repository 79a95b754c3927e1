"""Offline registration metrics and the per-frame result files of the reference's evaluation — row f4 of SURVEY.md §8.

    eval_all.py:121-131   one dict per frame saved as <eval_results>/<dataset>/%06d.npy      -> frame_result / save_frame_result
    calc_result.py:3-16   registration recall + mean/std of RRE, RTE under (r_thrs, t_thrs)  -> registration_recall / report_lines
    IR_RMSE.py:30-72      inlier ratio per pixel threshold 0..10 step 0.2 and RMSE per frame -> inlier_ratio_rmse / evaluate_result_files

Host-side numpy, like the reference's scripts (a few thousand points per frame: nothing here is worth a kernel).  Pinned against
the reference's own scripts run on synthetic result files: tests/tools/make_golden_metrics.py -> tests/golden/metrics.npz,
tests/test_metrics_cpu.py."""
import os
import warnings
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

FRAME_KEYS = ("GT_P", "pred_P", "K", "points", "P", "superpoints", "superpoints_score", "fine_xy", "object_points")
REPORT_THRESHOLDS = ((1e5, 1e5), (45, 10), (10, 5))   # (r_thrs, t_thrs) of calc_result.py:22-32


def _np(x) -> np.ndarray:
    """Result files hold torch tensors for everything the network produced (eval_all.py:123-130) and numpy for the poses."""
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# ------------------------------------------------------------------------------------------------ per-frame result files
def frame_result(GT_P, pred_P, K, points, superpoints, superpoints_score, fine_xy, object_points) -> Dict[str, object]:
    """The dict eval_all.py:121-130 saves for one frame.  `P` repeats GT_P (eval_all.py:125 stores the same variable twice).
    GT_P: the dataset's `P` entry (inverse of the random transform, kitti.py:380); pred_P: T_pred of eval_all.py:111-113;
    points: stage-1 points (10240,3); superpoints: last-stage points (1280,3); fine_xy (2,n); object_points (n,3)."""
    return {"GT_P": GT_P, "pred_P": pred_P, "K": K, "points": points, "P": GT_P, "superpoints": superpoints,
            "superpoints_score": superpoints_score, "fine_xy": fine_xy, "object_points": object_points}


def save_frame_result(directory: str, step: int, result: Dict[str, object]) -> str:
    """np.save(eval_path / '%06d.npy' % step, save_dict) of eval_all.py:131 (a pickled dict inside an .npy)."""
    missing = [k for k in FRAME_KEYS if k not in result]
    if missing:
        raise KeyError("frame result lacks %s" % missing)
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "%06d.npy" % step)
    np.save(path, result, allow_pickle=True)
    return path


def load_frame_result(path: str) -> Dict[str, object]:
    return np.load(path, allow_pickle=True).item()


# ------------------------------------------------------------------------------------------------ RRE / RTE recall
def registration_recall(r_error, t_error, r_thrs: float, t_thrs: float) -> Dict[str, float]:
    """calc_result.py:3-14: a frame succeeds when RRE < r_thrs and RTE < t_thrs (strict); mean / population std of both errors
    over the successful frames (NaN when none succeeds, as numpy's mean of an empty array)."""
    r, t = np.asarray(r_error), np.asarray(t_error)
    if r.shape != t.shape or r.ndim != 1:
        raise ValueError("r_error and t_error must be 1-D arrays of equal length")
    ok = (r < r_thrs) & (t < t_thrs)
    rs, ts = r[ok], t[ok]
    with warnings.catch_warnings():   # "mean of empty slice" when nothing succeeds: NaN, like the script
        warnings.simplefilter("ignore", RuntimeWarning)
        out = {"r_thrs": float(r_thrs), "t_thrs": float(t_thrs), "success_rate": float(ok.sum() / len(ok) * 100.0) if len(ok) else float("nan"),
               "r_mean": float(rs.mean()), "r_std": float(rs.std()), "t_mean": float(ts.mean()), "t_std": float(ts.std()),
               "num_success": int(ok.sum()), "num_frames": int(len(ok))}
    return out


def report_lines(stats: Dict[str, float]) -> List[str]:
    """The text calc_result.py:7-16 prints for one threshold pair."""
    return ["--------------error calculation---------------------",
            "r_thrs: %.2f, t_thrs: %.2f" % (stats["r_thrs"], stats["t_thrs"]),
            "rot thrs: %.4f, trans thrs: %.4f, successful rate %0.2f %%" % (stats["r_thrs"], stats["t_thrs"], stats["success_rate"]),
            "succ_r_mean: %.2f, succ_r_std: %.2f" % (stats["r_mean"], stats["r_std"]),
            "succ_t_mean: %.2f, succ_t_std: %.2f" % (stats["t_mean"], stats["t_std"]),
            "----------Done!----------"]


def report(r_error, t_error, thresholds: Sequence[Tuple[float, float]] = REPORT_THRESHOLDS) -> List[str]:
    """calc_result.py:19-32 as a function: the three threshold pairs of the paper's tables."""
    lines = []
    for r_thrs, t_thrs in thresholds:
        lines += report_lines(registration_recall(r_error, t_error, r_thrs, t_thrs))
    return lines


# ------------------------------------------------------------------------------------------------ inlier ratio / RMSE
def pixel_thresholds() -> np.ndarray:
    return np.arange(0, 10.2, 0.2)   # IR_RMSE.py:30


def gt_pixels(object_points, gt_P, K) -> np.ndarray:
    """IR_RMSE.py:49-51: object points projected with the ground-truth pose -> (2,n) pixels.  The script inverts GT_P and then
    applies the inverse of that (rotation inverted numerically, translation mapped through it); the same order of operations and
    the inputs' own dtypes (float32 when the files hold float32) are kept so the results agree to the last bits."""
    X = _np(object_points)
    P = np.linalg.inv(_np(gt_P))
    R_back = np.linalg.inv(P[0:3, 0:3])
    cam = R_back @ X.T - R_back @ P[0:3, 3:]
    proj = _np(K) @ cam
    return proj[0:2] / proj[2]


def inlier_ratio_rmse(fine_xy, object_points, gt_P, K, thresholds: Optional[np.ndarray] = None) -> Tuple[np.ndarray, float]:
    """One frame of IR_RMSE.py:36-58: residual_i = |fine_xy_i - gt_pixel_i|; IR(thr) = share of residuals <= thr; RMSE = mean
    residual (the reference's name for it).  -> (ir (T,), rmse)."""
    thr = pixel_thresholds() if thresholds is None else np.asarray(thresholds, dtype=np.float64)
    xy = _np(fine_xy)
    if xy.ndim != 2 or xy.shape[0] != 2:
        raise ValueError("fine_xy must be (2,n)")
    residual = np.sum(np.square(xy - gt_pixels(object_points, gt_P, K)), axis=0) ** 0.5
    if residual.shape[0] == 0:
        raise ValueError("frame without correspondences")
    ir = (residual[None, :] <= thr[:, None]).sum(1) / residual.shape[0]
    return ir, float(np.mean(residual))


def evaluate_result_files(paths: Iterable[str], thresholds: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """IR_RMSE.py:31-72 over result files (in the given order): -> (ir_thre_list (T,) = IR averaged over frames per threshold,
    rmse_thre_list (T, frames) = every frame's RMSE repeated per threshold, exactly what the script stores)."""
    thr = pixel_thresholds() if thresholds is None else np.asarray(thresholds, dtype=np.float64)
    irs, rmses = [], []
    for p in paths:
        d = load_frame_result(p)
        ir, rmse = inlier_ratio_rmse(d["fine_xy"], d["object_points"], d["GT_P"], d["K"], thr)
        irs.append(ir)
        rmses.append(rmse)
    if not irs:
        raise ValueError("no result files")
    irs = np.stack(irs, 1)                                   # (T, frames)
    return irs.mean(1), np.broadcast_to(np.asarray(rmses)[None, :], irs.shape).copy()
