"""The two stack-mode neighbourhood operators of model/kpconv/ops (grid_subsample.py, radius_search.py) on the GPU.

In the reference both forward to `geotransformer.ext`, an extension that is not vendored, and nothing on the forward path calls them
(the pipeline sub-samples at random and searches k nearest neighbours, preprocess_data.py:36-107).  They are built here because the
task names them, on kernels this repository already has: the stable-radix-sort voxel grid (csrc/dataside.hip) and the exact KNN
(csrc/knn.hip, csrc/knn_grid.hip).  Semantics follow the published C++ of the extension (KPConv-PyTorch / GeoTransformer
cpp_wrappers); parity with the extension itself is unpinned."""
from typing import Tuple

import torch

from . import _lib, ops


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_WS = ops.Workspace()


def grid_subsample(points: torch.Tensor, lengths: torch.Tensor, voxel_size: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """model/kpconv/ops/grid_subsample.py:7-22.  points (N,3) CUDA fp32 = B clouds stacked, lengths (B,) -> (s_points (M,3), s_lengths (B,)):
    the barycentre of every occupied cell of a `voxel_size` grid, cloud by cloud.  Cells come out in ascending (iz, iy, ix) order (the
    extension: hash-map order).  One host sync per cloud (its cell count)."""
    lib = _lib.load()
    if not points.is_cuda or points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3 or not points.is_contiguous():
        raise _lib.CofiError("grid_subsample: points must be contiguous CUDA float32 (N,3)")
    lens = [int(v) for v in lengths.tolist()]
    if sum(lens) != points.shape[0] or min(lens, default=1) < 0:
        raise _lib.CofiError("grid_subsample: lengths do not add up to the number of points")
    outs, counts, start = [], [], 0
    cnt = torch.zeros(2, dtype=torch.int32, device=points.device)
    for n in lens:
        if n == 0:
            counts.append(0)
            continue
        seg = points[start:start + n]
        out = torch.empty((n, 3), dtype=torch.float32, device=points.device)
        ws = _WS.get(lib.cofi_voxel_downsample_workspace(n), points.device)
        _lib.check(lib.cofi_grid_subsample(_p(seg), n, float(voxel_size), _p(out), n, _p(cnt), _p(ws), ws.numel(), _stream()), "cofi_grid_subsample")
        c = cnt.cpu()
        if int(c[1]):
            raise _lib.CofiError("grid_subsample: the cloud spans more than 8192 cells along an axis")
        outs.append(out[:int(c[0])])
        counts.append(int(c[0]))
        start += n
    s_points = torch.cat(outs) if outs else points.new_zeros((0, 3))
    return s_points, torch.tensor(counts, dtype=lengths.dtype, device=lengths.device)


def radius_search(q_points: torch.Tensor, s_points: torch.Tensor, q_lengths: torch.Tensor, s_lengths: torch.Tensor, radius: float,
                  neighbor_limit: int) -> torch.Tensor:
    """model/kpconv/ops/radius_search.py:7-27.  For every query the support points of ITS cloud closer than `radius`, nearest first, as
    indices into the STACKED support set; rows are filled with the total support count (the shadow index) and are as wide as the
    fullest row, at most `neighbor_limit` (1..128: the width of the exact k-nearest search underneath; the extension's unlimited
    mode, neighbor_limit <= 0, is not served).  Ties at equal distance: lowest index first (the extension: unspecified)."""
    lib = _lib.load()
    if not (0 < int(neighbor_limit) <= 128):
        raise _lib.CofiError("radius_search: neighbor_limit must be in 1..128")
    for t, name in ((q_points, "q_points"), (s_points, "s_points")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 3 or not t.is_contiguous():
            raise _lib.CofiError("radius_search: %s must be contiguous CUDA float32 (n,3)" % name)
    ql, sl = [int(v) for v in q_lengths.tolist()], [int(v) for v in s_lengths.tolist()]
    if len(ql) != len(sl) or sum(ql) != q_points.shape[0] or sum(sl) != s_points.shape[0]:
        raise _lib.CofiError("radius_search: lengths do not match the stacked point sets")
    k, total_s = int(neighbor_limit), s_points.shape[0]
    out = torch.empty((q_points.shape[0], k), dtype=torch.int64, device=q_points.device)
    maxc = torch.zeros(1, dtype=torch.int32, device=q_points.device)
    q0 = s0 = 0
    for nq, ns in zip(ql, sl):
        if nq:
            if ns == 0:
                out[q0:q0 + nq] = total_s
            else:
                sup, qry = s_points[s0:s0 + ns], q_points[q0:q0 + nq]
                grid = ops.KnnGrid(sup) if ns >= ops.KNN_GRID_MIN_SUPPORT else None
                idx, dist = ops.knn(sup, qry, k, return_dist=True, grid=grid)
                _lib.check(lib.cofi_radius_mask(_p(idx), _p(dist), nq, k, ns, float(radius), s0, total_s, _p(out[q0:q0 + nq]), _p(maxc), _stream()),
                           "cofi_radius_mask")
        q0, s0 = q0 + nq, s0 + ns
    width = int(maxc.item())
    return out[:, :width].contiguous() if width < k else out
