"""Point-cloud pyramid on the GPU: the HIP KNN kernel in the role of the reference's DataLoader-side
`precompute_point_cloud_stack_mode` (model/kpconv/preprocess_data.py:36-107; its importable torch twin
`precompute_point_cloud_cuda`, :145-203)."""
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .spec import NUM_NEIGHBORS


def morton_order(points: torch.Tensor) -> torch.Tensor:
    """int32 permutation that sorts the points along a 30-bit Morton (Z-order) curve.  Used only as the PROCESSING
    order of the gather kernels (cofi_kpconv_aggregate / cofi_neighbor_maxpool): consecutive waves then work on
    neighbouring queries whose KNN rows overlap, which turns L2 gathers into L1 hits.  Results do not depend on it."""
    lo, hi = points.min(0)[0], points.max(0)[0]
    q = ((points - lo) / (hi - lo).clamp_min(1e-9) * 1023.0).long().clamp_(0, 1023)

    def spread(v):  # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code).to(torch.int32).contiguous()


def build_pyramid(points: torch.Tensor, subsample: List[torch.Tensor], k: int = NUM_NEIGHBORS, int64: bool = False) -> Dict:
    """points (N,3) CUDA fp32; subsample[i] = indices (CUDA int64/int32) into stage i selecting stage
    i+1 (the reference draws them with np.random.choice WITH replacement, preprocess_data.py:58).
    Returns the reference's dict layout: neighbors[i] (N_i,k) into stage i, subsampling[i] (N_{i+1},k)
    into stage i, upsampling[i] (N_i,k) into stage i+1 — int32 (device native) or int64."""
    pts = [points.contiguous()]
    for sel in subsample:
        pts.append(pts[-1][sel.long()].contiguous())
    # one cell grid per large stage, shared by the (up to three) searches against it; queries are processed in their own
    # stage's cell order when it has one (neighbouring waves then read the same cells).  Small stages: brute force.
    grids = [ops.KnnGrid(p) if p.shape[0] >= ops.KNN_GRID_MIN_SUPPORT else None for p in pts]
    qord = [None if g is None else g.order for g in grids]
    neighbors, subsampling, upsampling = [], [], []
    for i in range(len(pts)):
        neighbors.append(ops.knn(pts[i], pts[i], k, grid=grids[i], qorder=qord[i] if grids[i] is not None else None))
        if i < len(pts) - 1:
            subsampling.append(ops.knn(pts[i], pts[i + 1], k, grid=grids[i], qorder=qord[i + 1] if grids[i] is not None else None))
            upsampling.append(ops.knn(pts[i + 1], pts[i], k, grid=grids[i + 1], qorder=qord[i] if grids[i + 1] is not None else None))
    conv = ops.idx_to_int64 if int64 else (lambda t: t)
    return {"points": pts, "lengths": [int(p.shape[0]) for p in pts], "neighbors": [conv(t) for t in neighbors],
            "subsampling": [conv(t) for t in subsampling], "upsampling": [conv(t) for t in upsampling],
            # optional extra key: a spatially coherent processing order for the gather kernels (results do not depend on it) —
            # the cell order of the stage's grid where there is one; the few thousand points of the small stages stay as they are
            "order": [g.order if g is not None else torch.arange(p.shape[0], dtype=torch.int32, device=p.device) for g, p in zip(grids, pts)]}


def precompute_point_cloud_stack_mode(points, intensity, normals, lengths, num_stages, device="cuda", rng: Optional[np.random.RandomState] = None):
    """Signature of preprocess_data.py:36.  points (3,N) numpy; intensity / normals are carried by the
    caller (kitti.py:293) and ignored here, exactly as in the reference."""
    import multiprocessing as mp

    if mp.current_process().name != "MainProcess" and mp.get_start_method(allow_none=True) in (None, "fork"):
        # the reference calls this inside Dataset.__getitem__ (data/kitti.py:292) under DataLoader workers: a FORKED worker cannot
        # initialise HIP (it hangs or fails inside the runtime) - say so instead
        raise RuntimeError("precompute_point_cloud_stack_mode launches HIP kernels and cannot run in a forked DataLoader worker: use "
                           "num_workers=0, or multiprocessing_context='spawn', or cofii2p_amd.dataside.FramePreparer (the device-side loader)")
    rng = np.random if rng is None else rng
    n = points.shape[1]
    sub = []
    for _ in range(num_stages - 1):
        sub.append(torch.from_numpy(rng.choice(np.arange(n), size=n // 2)).to(device))
        n //= 2
    p = torch.from_numpy(np.ascontiguousarray(points.T)).float().to(device)
    return build_pyramid(p, sub, int64=True)
