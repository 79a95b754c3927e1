"""Synthetic KITTI-shaped frames — there is no dataset access.

Host-side numpy only.  A frame mimics what `data/kitti.py:259-303` hands to the model:
a (3,H,W) image in [0,1), a voxel-lattice point cloud of exactly ``num_points`` points in the
camera frame (x right, y down, z forward) after a random yaw / in-plane shift
(`data/options.py:33-38`), and per-point features [intensity, nx, ny, nz] (`kitti.py:293`).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Frame:
    img: np.ndarray  # (3,H,W) float32
    points: np.ndarray  # (N,3) float32
    feats: np.ndarray  # (N,4) float32
    seed: int


def make_frame(frame_id: int = 0, num_points: int = 20480, img_hw=(160, 512), extent: float = 40.0) -> Frame:
    seed = 1234 + int(frame_id)
    g = np.random.default_rng(seed)
    H, W = img_hw
    img = g.random((3, H, W), dtype=np.float32)

    n_raw = int(num_points * 1.6)
    n_ground = int(0.7 * n_raw)
    n_fac = n_raw - n_ground
    # ground plane 1.65 m below the camera (y is down)
    gx = g.uniform(-extent, extent, n_ground)
    gz = g.uniform(-extent, extent, n_ground)
    gy = 1.65 + g.normal(0.0, 0.03, n_ground)
    ground = np.stack([gx, gy, gz], 1)
    g_normal = np.tile(np.array([0.0, -1.0, 0.0]), (n_ground, 1))
    # 8 vertical facades, each with x or z fixed
    fac_pts, fac_nrm = [], []
    per = n_fac // 8
    for f in range(8):
        cnt = per if f < 7 else n_fac - 7 * per
        fixed = g.uniform(-extent + 5, extent - 5)
        u = g.uniform(-extent, extent, cnt)
        h = g.uniform(-6.0, 1.65, cnt)
        if f % 2 == 0:
            fac_pts.append(np.stack([np.full(cnt, fixed), h, u], 1))
            fac_nrm.append(np.tile(np.array([1.0, 0.0, 0.0]), (cnt, 1)))
        else:
            fac_pts.append(np.stack([u, h, np.full(cnt, fixed)], 1))
            fac_nrm.append(np.tile(np.array([0.0, 0.0, 1.0]), (cnt, 1)))
    pts = np.concatenate([ground] + fac_pts, 0)
    nrm = np.concatenate([g_normal] + fac_nrm, 0)
    # voxel lattice 0.1 m + de-duplication (mimics voxel_down_sample(0.1), kitti.py:283)
    pts = np.round(pts / 0.1) * 0.1
    _, first = np.unique(np.round(pts / 0.1).astype(np.int64), axis=0, return_index=True)
    first.sort()
    pts, nrm = pts[first], nrm[first]
    # resample to exactly num_points (kitti.py:168-180 downsample_np)
    if pts.shape[0] >= num_points:
        pick = g.choice(pts.shape[0], num_points, replace=False)
    else:
        pick = np.concatenate([np.arange(pts.shape[0]), g.choice(pts.shape[0], num_points - pts.shape[0], replace=True)])
    pts, nrm = pts[pick], nrm[pick]
    # random yaw about y and x/z shift (options.py:33-38)
    yaw = g.uniform(0.0, 2 * np.pi)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    t = np.array([g.uniform(-10, 10), 0.0, g.uniform(-10, 10)])
    pts = pts @ R.T + t
    nrm = nrm @ R.T + g.normal(0.0, 0.05, nrm.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    inten = g.random((num_points, 1))
    feats = np.concatenate([inten, nrm], 1)
    return Frame(img, pts.astype(np.float32), feats.astype(np.float32), seed)


def subsample_indices(n: int, num_stages: int, seed: int):
    """Index lists of the random half sub-sampling WITH replacement
    (preprocess_data.py:55-59), stage i+1 from stage i."""
    g = np.random.RandomState(seed)
    out = []
    for _ in range(num_stages - 1):
        out.append(g.choice(np.arange(n), size=n // 2))
        n //= 2
    return out


# KITTI odometry calibration of sequence 09 / 10 as data/kitti.py:24-63 reads it from calib.txt ("P2: ..." 3x4 projection row,
# "Tr: ..." velodyne -> camera 0), rounded values of the public calibration files
KITTI_CALIB_LINES = {
    "P0": "7.070912000000e+02 0.000000000000e+00 6.018873000000e+02 0.000000000000e+00 0.000000000000e+00 7.070912000000e+02 1.831104000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00",
    "P2": "7.070912000000e+02 0.000000000000e+00 6.018873000000e+02 4.688783000000e+01 0.000000000000e+00 7.070912000000e+02 1.831104000000e+02 1.178601000000e-01 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 6.203223000000e-03",
    "P3": "7.070912000000e+02 0.000000000000e+00 6.018873000000e+02 -3.334597000000e+02 0.000000000000e+00 7.070912000000e+02 1.831104000000e+02 1.930130000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 3.318498000000e-03",
    "Tr": "-1.857739385241e-03 -9.999659513510e-01 -8.039975204516e-03 -4.784029760483e-03 -6.481465826011e-03 8.051860151134e-03 -9.999466081774e-01 -7.337429464231e-02 9.999773098287e-01 -1.805528627661e-03 -6.496203536139e-03 -3.339968064433e-01",
}


def make_raw_scan(frame_id: int = 0, num_points: int = 120000, img_hw=(376, 1241)):
    """A raw frame as it lies on disk for data/kitti.py:262-279: `data` (7, N) float32 = [xyz | intensity | normal] in the velodyne
    frame (x forward, y left, z up), `img` (H, W, 3) uint8, `K` (3, 3) float64.  Ground plane + facades sampled along rotating
    beams, denser near the sensor like a real scan, so the 0.1 m voxel grid merges several points per voxel close in and keeps
    single points far out."""
    g = np.random.default_rng(977 + int(frame_id))
    H, W = img_hw
    img = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
    n_ground = int(0.6 * num_points)
    # ground: range ~ 3..70 m with 1/r density, full azimuth
    r = np.exp(g.uniform(np.log(3.0), np.log(70.0), n_ground))
    az = g.uniform(-np.pi, np.pi, n_ground)
    ground = np.stack([r * np.cos(az), r * np.sin(az), -1.73 + g.normal(0.0, 0.02, n_ground)], 1)
    g_n = np.tile(np.array([0.0, 0.0, 1.0]), (n_ground, 1))
    pts, nrm = [ground], [g_n]
    n_fac = num_points - n_ground
    per = n_fac // 10
    for f in range(10):
        cnt = per if f < 9 else n_fac - 9 * per
        d = g.uniform(5.0, 45.0) * (1 if f % 4 < 2 else -1)
        u = g.uniform(-50.0, 50.0, cnt) * g.uniform(0.2, 1.0, cnt)
        h = g.uniform(-1.73, 4.0, cnt)
        if f % 2 == 0:
            pts.append(np.stack([np.full(cnt, d) + g.normal(0, 0.02, cnt), u, h], 1))
            nrm.append(np.tile(np.array([-np.sign(d), 0.0, 0.0]), (cnt, 1)))
        else:
            pts.append(np.stack([u, np.full(cnt, d) + g.normal(0, 0.02, cnt), h], 1))
            nrm.append(np.tile(np.array([0.0, -np.sign(d), 0.0]), (cnt, 1)))
    pts = np.concatenate(pts, 0)
    nrm = np.concatenate(nrm, 0) + g.normal(0.0, 0.05, (num_points, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    inten = g.random((num_points, 1)) * 0.99
    order = g.permutation(num_points)
    data = np.concatenate([pts, inten, nrm], 1)[order].T.astype(np.float32)
    K = np.array([[707.0912, 0.0, 601.8873], [0.0, 707.0912, 183.1104], [0.0, 0.0, 1.0]])
    return np.ascontiguousarray(data), img, K


def make_raw_nuscenes(frame_id: int = 0, num_points: int = 26000, img_hw=(900, 1600)):
    """A nuScenes-style sample as data/nuscenes.py:183-193 reads it: `pc` (4, N) float32 = [xyz | intensity] ALREADY in the camera
    frame (x right, y down, z forward), `img` (H, W, 3) uint8, `K` (3, 3) float64."""
    g = np.random.default_rng(4242 + int(frame_id))
    H, W = img_hw
    img = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
    n_ground = int(0.65 * num_points)
    r = np.exp(g.uniform(np.log(2.5), np.log(60.0), n_ground))
    az = g.uniform(-np.pi, np.pi, n_ground)
    ground = np.stack([r * np.sin(az), 1.8 + g.normal(0.0, 0.03, n_ground), r * np.cos(az)], 1)
    n_wall = num_points - n_ground
    u = g.uniform(-40.0, 40.0, n_wall)
    side = g.integers(0, 4, n_wall)
    d = g.uniform(8.0, 35.0, 4)[side]
    h = g.uniform(-5.0, 1.8, n_wall)
    wall = np.where((side % 2 == 0)[:, None], np.stack([np.where(side == 0, d, -d), h, u], 1), np.stack([u, h, np.where(side == 1, d, -d)], 1))
    pts = np.concatenate([ground, wall], 0)[g.permutation(num_points)]
    inten = g.uniform(0.0, 255.0, (num_points, 1))
    K = np.array([[1266.417, 0.0, 816.267], [0.0, 1266.417, 491.507], [0.0, 0.0, 1.0]])
    return np.ascontiguousarray(np.concatenate([pts, inten], 1).T.astype(np.float32)), img, K
