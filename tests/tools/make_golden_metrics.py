"""Golden vectors for cofii2p_amd/metrics.py: runs the REFERENCE's offline metric scripts (evaluation/calc_result.py,
evaluation/IR_RMSE.py — both are scripts, not modules, so they are executed with runpy inside a temporary working directory)
on seeded synthetic result files and stores the inputs and what the scripts produced in tests/golden/metrics.npz.

Development container only (needs /root/reference):   python tests/tools/make_golden_metrics.py
"""
import contextlib
import io
import os
import runpy
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402


def random_pose(rng):
    a = rng.uniform(0, 2 * np.pi)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = [rng.uniform(-10, 10), 0.0, rng.uniform(-10, 10)]
    return P


def synthetic_frame(rng, n):
    """A frame dict as eval_all.py:121-130 saves it: poses numpy, network outputs torch."""
    K = np.array([[360.0, 0, 300.0], [0, 360.0, 90.0], [0, 0, 1]], dtype=np.float32)     # KITTI P2 at 1/2 resolution, roughly
    gt_P = random_pose(rng).astype(np.float32)                                         # dataset 'P' = inv(random transform), float32
    cam = np.stack([rng.uniform(-8, 8, n), rng.uniform(-1, 2, n), rng.uniform(4, 40, n)])  # points in front of the camera
    obj = (np.linalg.inv(gt_P.astype(np.float64)) @ np.vstack([cam, np.ones(n)]))[:3].T  # so that GT_P maps them there
    pix = K.astype(np.float64) @ cam
    pix = pix[:2] / pix[2]
    noise = rng.normal(0, 1.0, (2, n)) * rng.choice([0.3, 2.0, 12.0], n)               # inliers, near misses, outliers
    fine_xy = np.floor(pix + noise).astype(np.float32)                                  # integer pixel coordinates like the network's
    pred = gt_P.astype(np.float64) @ random_pose(rng) * 1.0
    return {"GT_P": gt_P, "pred_P": pred, "K": torch.from_numpy(K), "points": torch.from_numpy(rng.normal(size=(64, 3)).astype(np.float32)),
            "P": gt_P, "superpoints": torch.from_numpy(rng.normal(size=(16, 3)).astype(np.float32)),
            "superpoints_score": torch.from_numpy(rng.uniform(size=(1, 1, 16)).astype(np.float32)),
            "fine_xy": torch.from_numpy(fine_xy), "object_points": torch.from_numpy(obj.astype(np.float32))}


def run_script(rel, argv, cwd):
    old_argv, old_cwd = sys.argv, os.getcwd()
    out = io.StringIO()
    sys.argv = [rel] + argv
    os.chdir(cwd)
    try:
        with contextlib.redirect_stdout(out):
            runpy.run_path(os.path.join(ref_shims.REF_ROOT, rel), run_name="__main__")
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    return out.getvalue()


def main():
    assert ref_shims.have_reference()
    sys.dont_write_bytecode = True
    ref_shims._stub_open3d()
    if ref_shims.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_shims.REF_ROOT)
    rng = np.random.default_rng(20260927)
    gold = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- IR_RMSE.py on 5 synthetic frames
        res_dir = os.path.join(tmp, "eval_results", "kitti")
        os.makedirs(res_dir)
        counts = [37, 4, 258, 120, 61]
        for i, n in enumerate(counts):
            f = synthetic_frame(rng, n)
            np.save(os.path.join(res_dir, "%06d.npy" % i), f, allow_pickle=True)
            for k in ("GT_P", "pred_P"):
                gold["f%d_%s" % (i, k)] = np.asarray(f[k])
            for k in ("K", "fine_xy", "object_points"):
                gold["f%d_%s" % (i, k)] = f[k].numpy()
        order = os.listdir(res_dir)                           # the order the script iterates in (same process, same directory)
        run_script("evaluation/IR_RMSE.py", ["kitti", "--eval_results_path", os.path.join(tmp, "eval_results")], tmp)
        gold["frame_order"] = np.array([int(os.path.splitext(f)[0]) for f in order])
        gold["ir"] = np.load(os.path.join(tmp, "cofii2p_kitti_ir_20480.npy"))
        gold["rmse"] = np.load(os.path.join(tmp, "cofii2p_kitti_rmse_20480.npy"))
        # ---- calc_result.py on synthetic error lists (it reads nuscenes_{r,t}_error.npy from the working directory)
        r = np.abs(rng.normal(0, 4, 300)) + rng.choice([0, 0, 0, 60], 300)
        t = np.abs(rng.normal(0, 1.5, 300)) + rng.choice([0, 0, 0, 12], 300)
        np.save(os.path.join(tmp, "nuscenes_r_error.npy"), r)
        np.save(os.path.join(tmp, "nuscenes_t_error.npy"), t)
        text = run_script("evaluation/calc_result.py", [], tmp)
        gold["r_error"], gold["t_error"] = r, t
        gold["calc_result_stdout"] = np.array(text)
    out = os.path.join(os.path.dirname(HERE), "golden", "metrics.npz")
    np.savez_compressed(out, **gold)
    print("wrote", out, os.path.getsize(out), "bytes; IR at 1/2/5 px:", gold["ir"][[5, 10, 25]])
    print(text)


if __name__ == "__main__":
    main()
