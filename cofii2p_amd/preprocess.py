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


def build_pyramid(points: torch.Tensor, subsample: List[torch.Tensor], k: int = NUM_NEIGHBORS, int64: bool = False, upsample_k: Optional[int] = None) -> Dict:
    """points (N,3) CUDA fp32; subsample[i] = indices (CUDA int64/int32) into stage i selecting stage
    i+1 (the reference draws them with np.random.choice WITH replacement, preprocess_data.py:58).
    Returns the reference's dict layout: neighbors[i] (N_i,k) into stage i, subsampling[i] (N_{i+1},k)
    into stage i, upsampling[i] (N_i,k) into stage i+1 — int32 (device native) or int64.
    upsample_k=1: upsampling[i] holds ONLY its first column, (N_i, 1) - the nearest stage-(i+1) point, all the forward reads of these
    tables (nearest_upsample, functional.py:20) - and is derived from neighbors[i] without a search (ops.knn_up_nearest: the same index
    the search returns, ties included).  The four up-searches are 48 % of a pyramid's kernel time; the forward's outputs do not change by
    a bit.  None / k: the reference's full (N_i, k) tables."""
    if upsample_k not in (None, 1, k):
        raise ValueError("upsample_k must be 1 (nearest only) or k (the reference's tables)")
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
            # subsampling[i] = the k nearest stage-i points of every stage-(i+1) point.  Stage i+1 is a SELECTION of stage i (point m is
            # stage-i point subsample[i][m], bit for bit), so that row is exactly row subsample[i][m] of neighbors[i] - same coordinates,
            # same canonical distances, same (distance, lowest index) order: 4 of the 13 searches of a pyramid are row gathers
            # (tests/test_ops_gpu.py::test_gpu_pyramid_bit_exact_and_int64_contract holds the tables to the searching oracle)
            subsampling.append(neighbors[i][subsample[i].long()].contiguous())
            if upsample_k == 1:
                upsampling.append(ops.knn_up_nearest(pts[i], neighbors[i], ops.idx_to_int32(subsample[i]).contiguous()))
            else:
                upsampling.append(ops.knn(pts[i + 1], pts[i], k, grid=grids[i + 1], qorder=qord[i] if grids[i + 1] is not None else None))
    conv = ops.idx_to_int64 if int64 else (lambda t: t)
    return {"points": pts, "lengths": [int(p.shape[0]) for p in pts], "neighbors": [conv(t) for t in neighbors],
            "subsampling": [conv(t) for t in subsampling], "upsampling": [conv(t) for t in upsampling],
            # optional extra key: a spatially coherent processing order for the gather kernels (results do not depend on it) —
            # the cell order of the stage's grid where there is one; the few thousand points of the small stages stay as they are
            "order": [g.order if g is not None else torch.arange(p.shape[0], dtype=torch.int32, device=p.device) for g, p in zip(grids, pts)]}


class PyramidGraph:
    """`build_pyramid` for clouds of ONE size as a hipGraph: the 5 grid builds and 13 KNN-128 searches of a frame are ~40 launches that a
    Python caller issues in ~0.9 ms of host time (more than the kernels take); captured once, a frame costs one batched input copy
    and one graph launch, and the tables come out in static int32 tensors a `forward_async(..., inputs_stable=True)` reads in place.
    One instance per frame in flight (its outputs are overwritten by the next run)."""

    def __init__(self, num_points: int, sub_sizes, device, capture_stream: Optional[torch.cuda.Stream] = None, k: int = NUM_NEIGHBORS,
                 upsample_k: Optional[int] = None):
        self.device = torch.device(device)
        self.upsample_k = upsample_k
        self.points = torch.empty((num_points, 3), dtype=torch.float32, device=self.device)
        self.sub = [torch.empty((n,), dtype=torch.int32, device=self.device) for n in sub_sizes]
        self.k, self.graph, self.out = k, None, None
        self._cap = capture_stream

    def run(self, points: torch.Tensor, sub: List[torch.Tensor]) -> Dict:
        """points (N,3) fp32, sub[i] int32 / int64 index tensors (device) -> the pyramid dict (static tensors, int32 tables)"""
        self.points.copy_(points, non_blocking=True)
        for d, s_ in zip(self.sub, sub):
            d.copy_(s_, non_blocking=True)   # converts int64 -> int32 on the way
        if self.graph is None:
            cap = self._cap or torch.cuda.Stream(device=self.device)
            cur = torch.cuda.current_stream()
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):
                for _ in range(2):
                    build_pyramid(self.points, self.sub, self.k, upsample_k=self.upsample_k)
            cur.wait_stream(cap)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=cap):
                self.out = build_pyramid(self.points, self.sub, self.k, upsample_k=self.upsample_k)
        self.graph.replay()
        return self.out


class FrameStack:
    """A stack-mode input (CoFiI2P.stack_frames layout: every tensor of B equally sized frames concatenated along its rows, index tables
    frame-local int32, images (B, 3, H, W)) that a loader fills FRAME BY FRAME: `put(f, pyramid, feats, img)` writes frame f's tensors
    into their row blocks with ONE batched copy launch (27 MB for a KITTI frame, ~10 us), so per-frame producers (PyramidGraph, FrameLoader
    slots) can feed `forward_async(slot, stack.pyr, stack.img, inputs_stable=True)` submissions of B frames without a gather pass or a
    host synchronisation.  The buffers are static: a ring of stacks = one hipGraph of the forward per stack."""
    KEYS = ("points", "neighbors", "subsampling", "upsampling")

    def __init__(self, pyramid: Dict, feats: torch.Tensor, img: torch.Tensor, B: int):
        dev = feats.device
        self.B = B
        self.pyr = {k: [torch.empty((B * t.shape[0],) + tuple(t.shape[1:]), dtype=torch.float32 if k == "points" else torch.int32, device=dev)
                        for t in pyramid[k]] for k in self.KEYS}
        self.pyr["feats"] = torch.empty((B * feats.shape[0], feats.shape[1]), dtype=torch.float32, device=dev)
        self.img = torch.empty((B,) + tuple(img.shape[-3:]), dtype=torch.float32, device=dev)
        self._mc = ops.MultiCopy(dev)
        self._mc.MAX_TABLES = 1 << 16   # one cached descriptor table per (producer slot, frame position): a few hundred bytes each

    def put(self, f: int, pyramid: Dict, feats: torch.Tensor, img: torch.Tensor):
        """frame f (0 <= f < B) <- the producer's tensors (int32 tables, contiguous), enqueued on the current stream"""
        if not 0 <= f < self.B:
            raise ValueError("frame position %d outside the stack of %d" % (f, self.B))
        srcs, dsts = [], []
        for k in self.KEYS:
            for t, d in zip(pyramid[k], self.pyr[k]):
                n = t.shape[0]
                srcs.append(t)
                dsts.append(d[f * n:(f + 1) * n])
        n = feats.shape[0]
        srcs += [feats, img.reshape(self.img.shape[1:])]
        dsts += [self.pyr["feats"][f * n:(f + 1) * n], self.img[f]]
        self._mc.run(srcs, dsts)


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
