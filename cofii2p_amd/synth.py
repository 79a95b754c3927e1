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
