"""cofii2p_amd/metrics.py against what the reference's offline scripts produced on the same files
(tests/golden/metrics.npz, recorded by tests/tools/make_golden_metrics.py from evaluation/IR_RMSE.py and evaluation/calc_result.py)."""
import os

import numpy as np
import pytest
import torch

from cofii2p_amd import metrics

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def _frames(gold):
    n = len(gold["frame_order"])
    return [{k: gold["f%d_%s" % (i, k)] for k in ("GT_P", "pred_P", "K", "fine_xy", "object_points")} for i in range(n)]


def test_inlier_ratio_and_rmse_match_reference_script(gold, tmp_path):
    frames = _frames(gold)
    paths = []
    for i in gold["frame_order"]:                                      # the order the script listed the directory in
        f = frames[int(i)]
        res = metrics.frame_result(f["GT_P"], f["pred_P"], torch.from_numpy(f["K"]), torch.zeros(8, 3), torch.zeros(4, 3), torch.zeros(1, 1, 4),
                                   torch.from_numpy(f["fine_xy"]), torch.from_numpy(f["object_points"]))
        paths.append(metrics.save_frame_result(str(tmp_path), int(i), res))
    ir, rmse = metrics.evaluate_result_files(paths)
    assert ir.shape == gold["ir"].shape == (51,) and rmse.shape == gold["rmse"].shape
    np.testing.assert_array_equal(ir, gold["ir"])                       # same operations on the same dtypes: bit-equal
    np.testing.assert_array_equal(rmse, gold["rmse"])
    assert ir[0] <= ir[-1] and np.all(np.diff(ir) >= 0)                 # IR is monotone in the pixel threshold


def test_frame_files_round_trip(gold, tmp_path):
    f = _frames(gold)[2]
    res = metrics.frame_result(f["GT_P"], f["pred_P"], f["K"], np.zeros((8, 3)), np.zeros((4, 3)), np.zeros((1, 1, 4)), f["fine_xy"], f["object_points"])
    assert tuple(res) == metrics.FRAME_KEYS                            # the schema of eval_all.py:121-130, in its order
    path = metrics.save_frame_result(str(tmp_path / "kitti"), 7, res)
    assert path.endswith("000007.npy")
    back = metrics.load_frame_result(path)
    for k in metrics.FRAME_KEYS:
        np.testing.assert_array_equal(np.asarray(back[k]), np.asarray(res[k]))
    with pytest.raises(KeyError):
        metrics.save_frame_result(str(tmp_path), 8, {"GT_P": f["GT_P"]})


def test_registration_report_matches_reference_script(gold):
    text = "\n".join(metrics.report(gold["r_error"], gold["t_error"])) + "\n"
    assert text == str(gold["calc_result_stdout"])
    s = metrics.registration_recall(gold["r_error"], gold["t_error"], 10, 5)
    ok = (gold["r_error"] < 10) & (gold["t_error"] < 5)
    assert s["num_success"] == int(ok.sum()) and s["num_frames"] == 300
    assert s["r_mean"] == float(gold["r_error"][ok].mean())


def test_edge_cases():
    none = metrics.registration_recall(np.array([50.0, 60.0]), np.array([1.0, 2.0]), 10, 5)     # nothing succeeds: NaN statistics, 0 %
    assert none["success_rate"] == 0.0 and np.isnan(none["r_mean"]) and np.isnan(none["t_std"])
    with pytest.raises(ValueError):
        metrics.registration_recall(np.zeros(3), np.zeros(4), 1, 1)
    with pytest.raises(ValueError):
        metrics.inlier_ratio_rmse(np.zeros((2, 0)), np.zeros((0, 3)), np.eye(4), np.eye(3))
    with pytest.raises(ValueError):
        metrics.evaluate_result_files([])
    # identity pose, points on the optical axis grid: residual exactly known
    K = np.array([[100.0, 0, 50], [0, 100.0, 40], [0, 0, 1]])
    X = np.array([[0.0, 0, 2], [1, 0, 2], [0, 1, 4]])
    xy = np.array([[50.0, 100.0, 50.0], [40.0, 40.0, 68.0]])           # exact, exact, 3 px off
    ir, rmse = metrics.inlier_ratio_rmse(xy, X, np.eye(4), K)
    assert rmse == pytest.approx(1.0) and ir[0] == pytest.approx(2 / 3) and ir[-1] == 1.0
