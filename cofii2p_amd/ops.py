"""Tensor-level host wrappers over the C ABI (torch is used for device memory + the stream only).

Every function takes torch CUDA tensors, validates layout, and enqueues the HIP kernels of
libcofi_hip.so on the CURRENT torch stream.  Row-major 2-D tensors may be column slices of a wider
buffer: the leading dimension is `stride(0)`, `stride(1)` must be 1.
"""
import ctypes
import math
from typing import Optional

import numpy as np
import torch

from . import _lib

import os

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
GEMM_BF16X3 = 0x100
# arithmetic of the dense contractions: "f32" = exact fp32 MFMA, "bf16x3" = 3-term bf16 split with fp32 accumulation
GEMM_MODE = os.environ.get("COFI_GEMM", "f32")


# Which intra-frame fork/join branches are taken (see Branch).  With >= 2 frames in flight the frames themselves fill
# the GPU and intra-frame forks only add join overhead; with one frame in flight they shorten the critical path.
BRANCH_MASK = int(os.environ.get("COFI_BRANCH_MASK", "7"))


def _gemm_flag() -> int:
    return GEMM_BF16X3 if GEMM_MODE == "bf16x3" else 0


GEMM_W_SPLIT = 0x200


class SplitW:
    """A static GEMM operand (weight) with its bf16 hi/lo planes, split ONCE (cofi_split_bf16_planes).  Accepted wherever a
    weight matrix is: the bf16x3 kernels read the planes (no on-the-fly conversion of W), the exact-fp32 kernels read `.w`."""

    def __init__(self, w: torch.Tensor):
        lib = _lib.load()
        _mat(w, "w")
        self.w = w
        self.shape, self.device, self.dtype = w.shape, w.device, w.dtype
        N, K = w.shape
        self.ldp = (K + 7) // 8 * 8
        self.planes = torch.empty((2, N, self.ldp), dtype=torch.int16, device=w.device)
        _lib.check(lib.cofi_split_bf16_planes(_p(w), _ld(w), N, K, _p(self.planes), self.ldp, _stream()), "cofi_split_bf16_planes")


    def numel(self):
        return self.w.numel()

    def dim(self):
        return 2


def presplit(w):
    return w if isinstance(w, SplitW) else SplitW(w)


def _wargs(w):
    """(pointer, leading dimension, extra flag) of a weight operand for the current GEMM mode."""
    if isinstance(w, SplitW):
        if GEMM_MODE == "bf16x3":
            return _p(w.planes), w.ldp, GEMM_W_SPLIT
        w = w.w
    return _p(w), _ld(w), 0


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _mat(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.CofiError("%s must be a CUDA (HIP) tensor — there is no CPU path" % name)
    if t.dtype != dtype or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise _lib.CofiError("%s: expected row-major 2-D %s, got %s %s stride %s" % (name, dtype, t.dtype, tuple(t.shape), t.stride()))
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _vec(t: Optional[torch.Tensor], name: str, n: int, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or t.numel() != n or not t.is_contiguous():
        raise _lib.CofiError("%s: expected contiguous CUDA %s of %d elements" % (name, dtype, n))
    return t


class Workspace:
    """Grow-only device scratch, one per (purpose, device, stream).  Kernels are stream ordered, so reuse by
    consecutive calls on ONE stream is safe; concurrent streams (forked branches of the forward graph) each
    get their own buffer.  Growing never frees the old buffer: a hipGraph captured earlier (another input shape on the same
    slot) keeps replaying with the address it recorded."""

    slot = 0  # frames-in-flight slot: concurrently replayed graphs must not share scratch (set_workspace_slot)

    def __init__(self):
        self.bufs = {}
        self.retired = []   # outgrown buffers stay alive: a captured hipGraph may still hold their address

    def get(self, nbytes: int, device) -> Optional[torch.Tensor]:
        if nbytes == 0:
            return None
        key = (device, torch.cuda.current_stream(device).cuda_stream, Workspace.slot)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                self.retired.append(buf)
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


class Branch:
    """Fork/join helper: run a branch of the forward on a side HIP stream (captured as a parallel branch
    when the forward is recorded into a hipGraph).

        with Branch(device, 0) as br:      # side stream waits for everything enqueued so far
            y = ops.gemm(...)              # enqueued on the side stream
        ...                                # main stream continues concurrently
        br.join(y)                         # main stream waits for the branch; y is safe to use
    """

    _pool = {}

    def __init__(self, device, slot: int = 0, enabled: bool = True):
        # BRANCH_MASK bit i enables fork/join slot i (0: image branch, 1: residual shortcut, 2: self-attention streams,
        # 3: ResNet layer3/4 tail that nothing downstream reads)
        self.enabled = enabled and bool((BRANCH_MASK >> slot) & 1)
        enabled = self.enabled
        self.device = device
        if enabled:
            key = (str(device), slot)
            if key not in Branch._pool:
                Branch._pool[key] = torch.cuda.Stream(device=device)
            self.side = Branch._pool[key]

    def __enter__(self):
        if self.enabled:
            self.main = torch.cuda.current_stream(self.device)
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self._ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.enabled:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.side)
            for t in tensors:
                if torch.is_tensor(t):
                    t.record_stream(cur)


_WS_GEMM = Workspace()
_WS_STATS = Workspace()


def set_workspace_slot(slot: int):
    """Select the scratch namespace used by subsequently enqueued / captured kernels (one per frame in flight)."""
    Workspace.slot = int(slot)


# ------------------------------------------------------------------------------------------ dense
def gemm(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, bias=None, rowdiv=None, act: int = ACT_NONE):
    """out[m,n] = act( (a @ w.T)[m,n] / rowdiv[m] + bias[n] );  a (M,K), w (N,K)."""
    lib = _lib.load()
    _mat(a, "a")
    if not isinstance(w, SplitW):
        _mat(w, "w")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.CofiError("gemm: K mismatch %s vs %s" % (tuple(a.shape), tuple(w.shape)))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _mat(out, "out")
    _vec(bias, "bias", N), _vec(rowdiv, "rowdiv", M)
    if M == 0:   # zero rows: nothing to launch (torch hands out a null pointer for an empty tensor, which the C ABI rejects)
        return out
    nbytes = lib.cofi_gemm_f32_workspace(M, N, K)
    ws = _WS_GEMM.get(nbytes, a.device)
    wp, wld, wflag = _wargs(w)
    rc = lib.cofi_gemm_f32(_p(a), _ld(a), wp, wld, _p(out), _ld(out), M, N, K, _p(bias), _p(rowdiv), act | _gemm_flag() | wflag, _p(ws),
                           0 if ws is None else ws.numel(), _stream())
    _lib.check(rc, "cofi_gemm_f32")
    return out


def gemm_colstats(a, w, out=None, bias=None, rowdiv=None, act: int = ACT_NONE):
    """gemm() that also returns the fused per-slab column statistics: (out, colpart (nslab,N,2))."""
    lib = _lib.load()
    _mat(a, "a")
    if not isinstance(w, SplitW):
        _mat(w, "w")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.CofiError("gemm: K mismatch %s vs %s" % (tuple(a.shape), tuple(w.shape)))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _mat(out, "out")
    _vec(bias, "bias", N), _vec(rowdiv, "rowdiv", M)
    nslab = lib.cofi_gemm_f32_stat_slabs(M, N, K)
    colpart = torch.empty((nslab, N, 2), dtype=torch.float32, device=a.device)
    ws = _WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, N, K), a.device)
    wp, wld, wflag = _wargs(w)
    rc = lib.cofi_gemm_f32_colstats(_p(a), _ld(a), wp, wld, _p(out), _ld(out), M, N, K, _p(bias), _p(rowdiv), act | _gemm_flag() | wflag, _p(colpart),
                                    _p(ws), 0 if ws is None else ws.numel(), _stream())
    _lib.check(rc, "cofi_gemm_f32_colstats")
    return out, colpart


def colstats_frames_ok(M_total: int, N: int, K: int, frames: int) -> bool:
    """True if the statistics slabs of a (M_total,N,K) contraction do not straddle frame boundaries (stack mode)."""
    if frames == 1:
        return True
    nslab = _lib.load().cofi_gemm_f32_stat_slabs(M_total, N, K)
    if nslab % frames or M_total % nslab:
        return False
    return (M_total // frames) % (M_total // nslab) == 0


def gemm_layernorm(a, w, gamma, beta, bias=None, relu: bool = False, res=None, out=None, eps: float = 1e-5):
    """out = relu?(LayerNorm(a @ w.T + bias) * gamma + beta) + res, one kernel (N <= 128)."""
    lib = _lib.load()
    _mat(a, "a")
    if not isinstance(w, SplitW):
        _mat(w, "w")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _mat(out, "out")
    ws = _WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, N, K), a.device)
    wp, wld, wflag = _wargs(w)
    rc = lib.cofi_gemm_f32_layernorm(_p(a), _ld(a), wp, wld, _p(out), _ld(out), M, N, K, _p(bias), _p(gamma), _p(beta), eps,
                                     int(relu) | _gemm_flag() | wflag, _p(res), 0 if res is None else _ld(res), _p(ws), 0 if ws is None else ws.numel(),
                                     _stream())
    _lib.check(rc, "cofi_gemm_f32_layernorm")
    return out


def group_stats_from_colpart(colpart, M: int, groups: int, eps: float = 1e-5, frames: int = 1):
    """M = total rows.  -> stats (groups,2), or (frames, groups, 2) in stack mode (slabs must not straddle frames)."""
    lib = _lib.load()
    nslab, C, _ = colpart.shape
    stats = torch.empty((groups, 2) if frames == 1 else (frames, groups, 2), dtype=torch.float32, device=colpart.device)
    _lib.check(lib.cofi_group_stats_from_colpart(_p(colpart), nslab, M, C, groups, eps, _p(stats), frames, _stream()),
               "cofi_group_stats_from_colpart")
    return stats


def col_inv_norm_from_colpart(colpart, C: int, eps: float = 1e-12, frames: int = 1):
    lib = _lib.load()
    nslab, ncols, _ = colpart.shape
    out = torch.empty((C,) if frames == 1 else (frames, C), dtype=torch.float32, device=colpart.device)
    _lib.check(lib.cofi_col_inv_norm_from_colpart(_p(colpart), nslab, ncols, C, eps, _p(out), frames, _stream()),
               "cofi_col_inv_norm_from_colpart")
    return out


# ------------------------------------------------------------------------------------------ KPConv
def row_sum_positive(feats: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _mat(feats, "feats")
    out = torch.empty((feats.shape[0],), dtype=torch.uint8, device=feats.device)
    _lib.check(lib.cofi_row_sum_positive(_p(feats), _ld(feats), feats.shape[0], feats.shape[1], _p(out), _stream()), "cofi_row_sum_positive")
    return out


def kpconv_aggregate(feats, q_pts, s_pts, idx, kernel_points, sigma: float, row_pos=None, frames: int = 1, order=None):
    """-> agg (M, 15*C), cnt (M,) float.  idx int32 (M,H).  Stack mode: `frames` equally sized frames stacked along
    the rows of every argument, idx frame-local."""
    lib = _lib.load()
    _mat(feats, "feats"), _mat(idx, "idx", torch.int32)
    if not (q_pts.is_contiguous() and s_pts.is_contiguous() and kernel_points.is_contiguous() and idx.is_contiguous()):
        raise _lib.CofiError("kpconv_aggregate: points / idx / kernel_points must be contiguous")
    N, C = feats.shape
    M, H = idx.shape
    if s_pts.shape != (N, 3) or q_pts.shape != (M, 3) or kernel_points.shape != (15, 3) or N % frames or M % frames:
        raise _lib.CofiError("kpconv_aggregate: shape mismatch")
    if row_pos is None:
        row_pos = getattr(feats, "cofi_row_pos", None)   # left there by the group_norm_apply that produced feats
    if row_pos is None:
        row_pos = row_sum_positive(feats)
    agg = torch.empty((M, 15 * C), dtype=torch.float32, device=feats.device)
    cnt = torch.empty((M,), dtype=torch.float32, device=feats.device)
    rc = lib.cofi_kpconv_aggregate(_p(feats), _ld(feats), N // frames, C, _p(q_pts), _p(s_pts), _p(idx), M // frames, H,
                                   _p(kernel_points), float(sigma), _p(row_pos), _p(agg), 15 * C, _p(cnt), frames, _p(order), _stream())
    _lib.check(rc, "cofi_kpconv_aggregate")
    return agg, cnt


def neighbor_maxpool(x, idx, out=None, frames: int = 1, order=None):
    lib = _lib.load()
    _mat(x, "x"), _mat(idx, "idx", torch.int32)
    M, H = idx.shape
    if out is None:
        out = torch.empty((M, x.shape[1]), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    _lib.check(lib.cofi_neighbor_maxpool(_p(x), _ld(x), x.shape[0] // frames, x.shape[1], _p(idx), M // frames, H, _p(out), _ld(out),
                                         frames, _p(order), _stream()), "cofi_neighbor_maxpool")
    return out


def gather_rows(x, idx, out=None, frames: int = 1):
    """out[m] = x[idx[m, 0]] (zero row for idx == N).  idx (M,) or (M,H) int32."""
    lib = _lib.load()
    _mat(x, "x")
    if idx.dtype != torch.int32 or not idx.is_cuda:
        raise _lib.CofiError("gather_rows: idx must be CUDA int32")
    M = idx.shape[0]
    stride = idx.stride(0) if idx.dim() == 2 else 1
    if out is None:
        out = torch.empty((M, x.shape[1]), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    _lib.check(lib.cofi_gather_rows(_p(x), _ld(x), x.shape[0] // frames, x.shape[1], _p(idx), stride, M // frames, _p(out), _ld(out),
                                    frames, _stream()), "cofi_gather_rows")
    return out


# ------------------------------------------------------------------------------------------ norms
def group_stats(x, groups: int, eps: float = 1e-5, frames: int = 1):
    """-> stats (groups,2), or (frames, groups, 2) in stack mode (x = frames blocks of M/frames rows)."""
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    stats = torch.empty((groups, 2) if frames == 1 else (frames, groups, 2), dtype=torch.float32, device=x.device)
    ws = _WS_STATS.get(lib.cofi_group_stats_workspace(M, C, groups, frames), x.device)
    _lib.check(lib.cofi_group_stats(_p(x), _ld(x), M, C, groups, eps, _p(stats), _p(ws), ws.numel(), frames, _stream()), "cofi_group_stats")
    return stats


class ColStats:
    """GroupNorm / InstanceNorm statistics still in the form the GEMM epilogue left them: per-slab column partials
    (nslab, C, 2) of an (M, C) activation.  `group_norm_apply` folds small tables inside its own kernel
    (cofi_group_norm_apply_colpart); large ones are finalised by cofi_group_stats_from_colpart first."""
    FUSE_MAX_PARTIALS = int(os.environ.get("COFI_GN_FUSE_MAX", "2048"))  # per frame: nslab * C partial pairs = 8 loads per thread, one L2 round trip

    def __init__(self, part: torch.Tensor, M: int, groups: int, frames: int = 1, eps: float = 1e-5):
        self.part, self.M, self.groups, self.frames, self.eps = part, M, groups, frames, eps

    def fusable(self) -> bool:
        nslab, C, _ = self.part.shape
        return (C & (C - 1)) == 0 and C <= 1024 and C // self.groups <= 256 and (nslab // self.frames) * C <= self.FUSE_MAX_PARTIALS

    def finalize(self) -> torch.Tensor:
        return group_stats_from_colpart(self.part, self.M, self.groups, self.eps, self.frames)


def group_norm_apply(x, stats, gamma=None, beta=None, slope: float = 1.0, res=None, res_stats=None, res_gamma=None, res_beta=None,
                     out=None, frames: int = 1, want_row_pos: bool = False):
    """stats: (groups,2) / stack mode (frames, groups, 2) tensor, or ColStats (column partials, folded in-kernel when small).
    want_row_pos (activation at most 256 wide): the kernel also emits row_pos = (row sum > 0), attached to the result as
    `out.cofi_row_pos` for the KPConv that consumes it (kpconv.py:113-114)."""
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
    c4n = C // 4
    row_pos = torch.empty((M,), dtype=torch.uint8, device=x.device) if (want_row_pos and C % 4 == 0 and c4n <= 64 and 64 % c4n == 0) else None
    fused = isinstance(stats, ColStats) and stats.fusable() and (res_stats is None or (isinstance(res_stats, ColStats) and res_stats.fusable()))
    if fused:
        rc = lib.cofi_group_norm_apply_colpart(_p(x), _ld(x), M, C, stats.groups, _p(stats.part), stats.part.shape[0], stats.eps, _p(gamma),
                                               _p(beta), _p(res), 0 if res is None else _ld(res),
                                               None if res_stats is None else _p(res_stats.part),
                                               0 if res_stats is None else res_stats.part.shape[0], _p(res_gamma), _p(res_beta),
                                               float(slope), _p(out), _ld(out), _p(row_pos), frames, _stream())
        _lib.check(rc, "cofi_group_norm_apply_colpart")
    else:
        if isinstance(stats, ColStats):
            stats = stats.finalize()
        if isinstance(res_stats, ColStats):
            res_stats = res_stats.finalize()
        groups = stats.shape[-2]
        rc = lib.cofi_group_norm_apply(_p(x), _ld(x), M, C, groups, _p(stats), _p(gamma), _p(beta), _p(res), 0 if res is None else _ld(res),
                                       _p(res_stats), _p(res_gamma), _p(res_beta), float(slope), _p(out), _ld(out), _p(row_pos), frames,
                                       _stream())
        _lib.check(rc, "cofi_group_norm_apply")
    if row_pos is not None:
        out.cofi_row_pos = row_pos
    return out


def group_norm(x, groups, gamma=None, beta=None, slope=1.0, eps=1e-5, out=None):
    return group_norm_apply(x, group_stats(x, groups, eps), gamma, beta, slope, out=out)


def layer_norm(x, gamma, beta, relu: bool = False, res=None, out=None, eps: float = 1e-5):
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    rc = lib.cofi_layer_norm(_p(x), _ld(x), M, C, _p(gamma), _p(beta), eps, int(relu), _p(res), 0 if res is None else _ld(res), _p(out),
                             _ld(out), _stream())
    _lib.check(rc, "cofi_layer_norm")
    return out


def col_inv_norm(x, eps: float = 1e-12):
    lib = _lib.load()
    _mat(x, "x")
    out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    _lib.check(lib.cofi_col_inv_norm(_p(x), _ld(x), x.shape[0], x.shape[1], eps, _p(out), _stream()), "cofi_col_inv_norm")
    return out


def l2norm_rows(x, out=None, transpose: bool = False):
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    if out is None:
        out = torch.empty((C, M) if transpose else (M, C), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    _lib.check(lib.cofi_l2norm_rows(_p(x), _ld(x), M, C, _p(out), _ld(out), int(transpose), _stream()), "cofi_l2norm_rows")
    return out


def l2norm_cols(x_cp, want_map: bool = True, want_tokens: bool = True, tokens_out=None):
    """x (C,P) channel-major -> (normalised (C,P) map, token-major (P,C) copy)."""
    lib = _lib.load()
    _mat(x_cp, "x")
    C, P = x_cp.shape
    y_cp = torch.empty((C, P), dtype=torch.float32, device=x_cp.device) if want_map else None
    y_pc = tokens_out if tokens_out is not None else (torch.empty((P, C), dtype=torch.float32, device=x_cp.device) if want_tokens else None)
    rc = lib.cofi_l2norm_cols(_p(x_cp), _ld(x_cp), C, P, _p(y_cp), 0 if y_cp is None else _ld(y_cp), _p(y_pc),
                              0 if y_pc is None else _ld(y_pc), _stream())
    _lib.check(rc, "cofi_l2norm_cols")
    return y_cp, y_pc


def transpose(x, out=None):
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    if out is None:
        out = torch.empty((C, M), dtype=torch.float32, device=x.device)
    _lib.check(lib.cofi_transpose(_p(x), _ld(x), M, C, _p(out), _ld(out), _stream()), "cofi_transpose")
    return out


def sine_frequencies(n_dim: int, d_model: int = 128, temperature: float = 10000.0) -> np.ndarray:
    """position_encoding.py:39-40 evaluated with the same fp32 torch ops (host constant table)."""
    f = d_model // n_dim // 2 * 2
    i = torch.arange(f, dtype=torch.float32)
    return (temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / f)).numpy().astype(np.float32)


def pos_sine(coords, out, accumulate: bool, d_model: int = 128):
    """out (T, >=d_model) (+)= PositionEmbeddingCoordsSine(coords).  coords float32 (T,n) or int32 grid."""
    lib = _lib.load()
    if not coords.is_cuda or not coords.is_contiguous() or coords.dtype not in (torch.float32, torch.int32):
        raise _lib.CofiError("pos_sine: coords must be contiguous CUDA float32/int32")
    T, n_dim = coords.shape
    dim_t = sine_frequencies(n_dim, d_model)
    rc = lib.cofi_pos_sine(_p(coords), int(coords.dtype == torch.int32), T, n_dim, dim_t.ctypes.data_as(ctypes.c_void_p), len(dim_t),
                           d_model, int(accumulate), _p(out), _ld(out), _stream())
    _lib.check(rc, "cofi_pos_sine")
    return out


# ------------------------------------------------------------------------------------------ image branch, NHWC
def conv2d_nhwc(x, H: int, W: int, w, ks: int, stride: int = 1, pad: int = 1, bias=None, res=None, act: int = ACT_NONE,
                colstats: bool = False, out=None, frames: int = 1):
    """Implicit-GEMM convolution on an NHWC map x (H*W, Cin) [row-major view, any leading dimension];
    w (Cout, ks*ks*Cin).  -> y (Ho*Wo, Cout) [, colpart]."""
    lib = _lib.load()
    _mat(x, "x")
    if not isinstance(w, SplitW):
        _mat(w, "w")
    Cin, Cout = x.shape[1], w.shape[0]
    if x.shape[0] != frames * H * W or w.shape[1] != ks * ks * Cin:
        raise _lib.CofiError("conv2d_nhwc: shape mismatch x %s w %s H %d W %d ks %d" % (tuple(x.shape), tuple(w.shape), H, W, ks))
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    M, K = frames * Ho * Wo, ks * ks * Cin
    if out is None:
        out = torch.empty((M, Cout), dtype=torch.float32, device=x.device)
    part = None
    if colstats:
        part = torch.empty((lib.cofi_gemm_f32_stat_slabs(M, Cout, K), Cout, 2), dtype=torch.float32, device=x.device)
    ws = _WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, Cout, K), x.device)
    wp, _wld, wflag = _wargs(w)   # convolution weights are dense (Cout, K) / planes (2, Cout, roundup8(K))
    rc = lib.cofi_conv2d_nhwc(_p(x), _ld(x), H, W, Cin, wp, Cout, ks, stride, pad, _p(bias), _p(res), 0 if res is None else _ld(res),
                              act | _gemm_flag() | wflag, _p(out), _ld(out), _p(part), _p(ws), 0 if ws is None else ws.numel(), frames, _stream())
    _lib.check(rc, "cofi_conv2d_nhwc")
    return (out, part, Ho, Wo) if colstats else (out, Ho, Wo)


def im2col_stem(img, kpad: int = 160):
    """img (3,H,W) or (frames,3,H,W) contiguous -> (frames*Ho*Wo, kpad), Ho, Wo."""
    lib = _lib.load()
    if img.dim() == 3:
        img = img[None]
    F_, C, H, W = img.shape
    if C != 3 or not img.is_contiguous():
        raise _lib.CofiError("im2col_stem: contiguous (frames,3,H,W) image expected")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((F_ * Ho * Wo, kpad), dtype=torch.float32, device=img.device)
    _lib.check(lib.cofi_im2col_stem(_p(img), H, W, kpad, _p(out), F_, _stream()), "cofi_im2col_stem")
    return out, Ho, Wo


def maxpool3x3s2_nhwc(x, H: int, W: int, frames: int = 1):
    lib = _lib.load()
    C = x.shape[1]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((frames * Ho * Wo, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.cofi_maxpool3x3s2_nhwc(_p(x), H, W, C, _p(out), frames, _stream()), "cofi_maxpool3x3s2_nhwc")
    return out, Ho, Wo


def upsample2x_cat_nhwc(low, h: int, w: int, skip, frames: int = 1):
    """low (frames*h*w, C1), skip (frames*4*h*w, C2) -> (frames*4*h*w, C1+C2)."""
    lib = _lib.load()
    _mat(low, "low"), _mat(skip, "skip")
    C1, C2 = low.shape[1], skip.shape[1]
    out = torch.empty((frames * 4 * h * w, C1 + C2), dtype=torch.float32, device=low.device)
    _lib.check(lib.cofi_upsample2x_cat_nhwc(_p(low), _ld(low), C1, h, w, _p(skip), _ld(skip), C2, _p(out), _ld(out), frames, _stream()),
               "cofi_upsample2x_cat_nhwc")
    return out


def extract_patches_nhwc(fmap, H2: int, W2: int, xy, cnt, cap: int, center_scale: float):
    lib = _lib.load()
    _mat(fmap, "fmap")
    C = fmap.shape[1]
    out = torch.empty((cap, C, 16), dtype=torch.float32, device=fmap.device)
    rc = lib.cofi_extract_patches_nhwc(_p(fmap), _ld(fmap), C, H2, W2, _p(xy), xy.stride(0), float(center_scale), _p(cnt), cap, _p(out),
                                       _stream())
    _lib.check(rc, "cofi_extract_patches_nhwc")
    return out


# ------------------------------------------------------------------------------------------ image-branch glue (NCHW / MIOpen variant)
def instance_norm_nchw(x, relu: bool = False, res=None, res_norm: bool = False, eps: float = 1e-5):
    """x (1,C,H,W) contiguous -> relu?(IN(x) + [res | IN(res)])."""
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[0] != 1:
        raise _lib.CofiError("instance_norm_nchw: contiguous CUDA fp32 (1,C,H,W) expected")
    _, C, H, W = x.shape
    y = torch.empty_like(x)
    mode = 0 if res is None else (2 if res_norm else 1)
    _lib.check(lib.cofi_instance_norm_nchw(_p(x), C, H * W, eps, _p(res), mode, int(relu), _p(y), _stream()), "cofi_instance_norm_nchw")
    return y


def bias_act_nchw(x, bias=None, res=None, res_bias=None, relu: bool = True):
    lib = _lib.load()
    _, C, H, W = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.cofi_bias_act_nchw(_p(x), _p(bias), _p(res), _p(res_bias), C, H * W, int(relu), _p(y), _stream()), "cofi_bias_act_nchw")
    return y


def upsample2x_cat(low, skip):
    """(1,C1,h,w), (1,C2,2h,2w) -> (1,C1+C2,2h,2w): bilinear x2 (align_corners=False) + channel concat."""
    lib = _lib.load()
    _, C1, h, w = low.shape
    C2 = skip.shape[1]
    out = torch.empty((1, C1 + C2, 2 * h, 2 * w), dtype=torch.float32, device=low.device)
    _lib.check(lib.cofi_upsample2x_cat(_p(low.contiguous()), C1, h, w, _p(skip.contiguous()), C2, _p(out), _stream()), "cofi_upsample2x_cat")
    return out


# ------------------------------------------------------------------------------------------ attention
def attention(q, k, v, q_colscale=None, nhead: int = 4, out=None, frames: int = 1, q_colpart=None, q_eps: float = 1e-12):
    """Stack mode: q (frames*L, HD), k/v (frames*S, HD), q_colscale (frames, HD).  q_colpart (nslab, ncols, 2): the column
    partials of the GEMM that produced q (its first HD columns) - the token-axis norm of Q is then folded inside the kernel."""
    lib = _lib.load()
    _mat(q, "q"), _mat(k, "k"), _mat(v, "v")
    L, HD = q.shape
    S = k.shape[0]
    D = HD // nhead
    if out is None:
        out = torch.empty((L, HD), dtype=torch.float32, device=q.device)
    if q_colpart is not None:
        rc = lib.cofi_attention_fwd_colpart(_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(q_colpart), q_colpart.shape[0], q_colpart.shape[1],
                                            q_eps, _p(out), _ld(out), L // frames, S // frames, nhead, D, 1.0 / math.sqrt(D), frames, _stream())
        _lib.check(rc, "cofi_attention_fwd_colpart")
        return out
    rc = lib.cofi_attention_fwd(_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(q_colscale), _p(out), _ld(out), L // frames, S // frames,
                                nhead, D, 1.0 / math.sqrt(D), None, 0, frames, _stream())
    _lib.check(rc, "cofi_attention_fwd")
    return out


def split_bf16(w: torch.Tensor):
    """fp32 -> (hi, lo) bf16 planes as int16 tensors: hi = bf16(w) (RNE), lo = bf16(w - hi)."""
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    return hi.contiguous().view(torch.int16), lo.contiguous().view(torch.int16)


def loftr_tail(msg, x, w, out, eps: float = 1e-5):
    """out = x + LN2(relu([x | LN1(msg Wm^T)] W0^T) W2^T) in one kernel; `w` holds the pre-split planes."""
    lib = _lib.load()
    _mat(msg, "msg"), _mat(x, "x"), _mat(out, "out")
    L = msg.shape[0]
    rc = lib.cofi_loftr_tail_bf16x3(_p(msg), _ld(msg), _p(x), _ld(x), _p(w["merge.hi"]), _p(w["merge.lo"]), _p(w["norm1.weight"]),
                                    _p(w["norm1.bias"]), _p(w["mlp.0.hi"]), _p(w["mlp.0.lo"]), _p(w["mlp.2.hi"]), _p(w["mlp.2.lo"]),
                                    _p(w["norm2.weight"]), _p(w["norm2.bias"]), eps, _p(out), _ld(out), L, _stream())
    _lib.check(rc, "cofi_loftr_tail_bf16x3")
    return out


class MultiCopy:
    """One-launch copy of a list of (src -> dst) tensor pairs (cofi_multi_copy).  The descriptor table of an address set is
    built once (pinned host buffer -> device) and cached: a stream of frames that recycles its buffers pays one kernel launch."""
    MAX_TABLES = 32

    def __init__(self, device):
        self.device = device
        self.tables = {}

    def run(self, srcs, dsts):
        lib = _lib.load()
        pairs = [(s, d) for s, d in zip(srcs, dsts) if s is not None]
        if not pairs:
            return
        key = tuple((s.data_ptr(), d.data_ptr(), s.numel() * s.element_size()) for s, d in pairs)
        table = self.tables.get(key)
        if table is None:
            for s, d in pairs:
                if not (s.is_contiguous() and d.is_contiguous() and s.shape == d.shape and s.dtype == d.dtype and s.is_cuda and d.is_cuda):
                    raise _lib.CofiError("multi_copy: contiguous CUDA tensors of equal shape / dtype expected")
            host = torch.tensor([v for k in key for v in k], dtype=torch.int64).pin_memory()
            table = torch.empty(3 * len(pairs), dtype=torch.int64, device=self.device)
            table.copy_(host, non_blocking=True)   # the pinned allocator keeps `host` alive until the copy has run
            if len(self.tables) >= self.MAX_TABLES:
                self.tables.pop(next(iter(self.tables)))
            self.tables[key] = table
        _lib.check(lib.cofi_multi_copy(_p(table), len(pairs), 64, _stream()), "cofi_multi_copy")


# ------------------------------------------------------------------------------------------ KNN / indices
KNN_GRID_MIN_SUPPORT = int(os.environ.get("COFI_KNN_GRID_MIN", "1024"))   # smaller support sets: brute force


def _check_xyz(t, name):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 3 or not t.is_contiguous():
        raise _lib.CofiError("knn: %s must be contiguous CUDA float32 (n,3)" % name)


class KnnGrid:
    """Cell grid over one support set (cofi_knn_grid_build): build once, search it with any number of query sets.
    `order` = the support indices in cell order (int32), a spatially coherent processing order for a self search."""

    def __init__(self, support: torch.Tensor, want_order: bool = True):
        lib = _lib.load()
        _check_xyz(support, "support")
        self.support, self.S = support, support.shape[0]
        self.ws = torch.empty(lib.cofi_knn_grid_workspace(self.S), dtype=torch.uint8, device=support.device)
        self.order = torch.empty((self.S,), dtype=torch.int32, device=support.device) if want_order else None
        _lib.check(lib.cofi_knn_grid_build(_p(support), self.S, _p(self.ws), self.ws.numel(), _p(self.order), _stream()), "cofi_knn_grid_build")

    def search(self, query: torch.Tensor, k: int, return_dist: bool = False, qorder: Optional[torch.Tensor] = None):
        lib = _lib.load()
        _check_xyz(query, "query")
        Q = query.shape[0]
        if qorder is not None and (qorder.dtype != torch.int32 or qorder.numel() != Q or not qorder.is_contiguous() or not qorder.is_cuda):
            raise _lib.CofiError("knn: qorder must be a contiguous CUDA int32 permutation of the queries")
        idx = torch.empty((Q, k), dtype=torch.int32, device=query.device)
        dist = torch.empty((Q, k), dtype=torch.float32, device=query.device) if return_dist else None
        if Q == 0:
            return (idx, dist) if return_dist else idx
        _lib.check(lib.cofi_knn_topk_grid(_p(self.ws), self.ws.numel(), self.S, _p(query), _p(qorder), Q, k, _p(idx), _p(dist), _stream()),
                   "cofi_knn_topk_grid")
        return (idx, dist) if return_dist else idx


def knn(support, query, k: int, return_dist: bool = False, grid=None, qorder=None):
    """k nearest support rows per query row, ascending (distance, index).  grid: a KnnGrid of `support` (same results, fewer
    distance evaluations); None = brute force (cofi_knn_topk)."""
    if grid is not None:
        if grid.support is not support and (grid.S != support.shape[0] or grid.support.data_ptr() != support.data_ptr()):
            raise _lib.CofiError("knn: the grid was built over another support set")
        return grid.search(query, k, return_dist, qorder)
    lib = _lib.load()
    _check_xyz(support, "support")
    _check_xyz(query, "query")
    Q = query.shape[0]
    idx = torch.empty((Q, k), dtype=torch.int32, device=query.device)
    dist = torch.empty((Q, k), dtype=torch.float32, device=query.device) if return_dist else None
    if Q == 0:
        return (idx, dist) if return_dist else idx
    if support.shape[0] == 0:
        raise _lib.CofiError("knn: empty support set")
    _lib.check(lib.cofi_knn_topk(_p(support), support.shape[0], _p(query), Q, k, _p(idx), _p(dist), _stream()), "cofi_knn_topk")
    return (idx, dist) if return_dist else idx


def nearest_node(nodes, points):
    lib = _lib.load()
    out = torch.empty((points.shape[0],), dtype=torch.int32, device=points.device)
    if points.shape[0] == 0:
        return out
    if nodes.shape[0] == 0:
        raise _lib.CofiError("nearest_node: empty node set")
    _lib.check(lib.cofi_nearest_node(_p(nodes.contiguous()), nodes.shape[0], _p(points.contiguous()), points.shape[0], _p(out), _stream()),
               "cofi_nearest_node")
    return out


def idx_to_int32(idx64: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if idx64.dtype == torch.int32:
        return idx64
    if idx64.dtype != torch.int64 or not idx64.is_cuda:
        raise _lib.CofiError("idx_to_int32: expected CUDA int64")
    src = idx64.contiguous()
    out = torch.empty(src.shape, dtype=torch.int32, device=src.device)
    _lib.check(lib.cofi_idx64_to_idx32(_p(src), _p(out), src.numel(), _stream()), "cofi_idx64_to_idx32")
    return out


def idx_to_int64(idx32: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    src = idx32.contiguous()
    out = torch.empty(src.shape, dtype=torch.int64, device=src.device)
    _lib.check(lib.cofi_idx32_to_idx64(_p(src), _p(out), src.numel(), _stream()), "cofi_idx32_to_idx64")
    return out


# ------------------------------------------------------------------------------------------ matching
def row_argmin_1m(sim):
    lib = _lib.load()
    _mat(sim, "sim")
    out = torch.empty((sim.shape[0],), dtype=torch.int32, device=sim.device)
    _lib.check(lib.cofi_row_argmin_1m(_p(sim), _ld(sim), sim.shape[0], sim.shape[1], _p(out), _stream()), "cofi_row_argmin_1m")
    return out


def select_matches(score, pix, W8: int, H8: int, thresholds: np.ndarray, min_matches: int = 4):
    """-> sel (N,) int32, coarse_xy (2,N) float32, count_dev (2,) int32 [n, threshold index]."""
    lib = _lib.load()
    N = score.numel()
    sel = torch.empty((N,), dtype=torch.int32, device=score.device)
    xy = torch.empty((2, N), dtype=torch.float32, device=score.device)
    cnt = torch.empty((2,), dtype=torch.int32, device=score.device)
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    rc = lib.cofi_select_matches(_p(score), _p(pix), N, W8, H8, thr.ctypes.data_as(ctypes.c_void_p), len(thr), min_matches, _p(sel),
                                 _p(xy), _p(cnt), _stream())
    _lib.check(rc, "cofi_select_matches")
    return sel, xy, cnt


def gather_points_sel(pts, sel, cnt):
    lib = _lib.load()
    cap = sel.numel()
    out = torch.empty((cap, 3), dtype=torch.float32, device=pts.device)
    _lib.check(lib.cofi_gather_points_sel(_p(pts), _p(sel), _p(cnt), cap, _p(out), _stream()), "cofi_gather_points_sel")
    return out


def nearest_node_sel(nodes, points_all, sel, cnt):
    lib = _lib.load()
    cap = sel.numel()
    out = torch.empty((cap,), dtype=torch.int32, device=nodes.device)
    rc = lib.cofi_nearest_node_sel(_p(nodes), nodes.shape[0], _p(points_all), _p(sel), _p(cnt), cap, _p(out), _stream())
    _lib.check(rc, "cofi_nearest_node_sel")
    return out


def extract_patches(fmap_chw, xy, cnt, cap: int, center_scale: float):
    lib = _lib.load()
    C, H2, W2 = fmap_chw.shape
    out = torch.empty((cap, C, 16), dtype=torch.float32, device=fmap_chw.device)
    rc = lib.cofi_extract_patches(_p(fmap_chw), C, H2, W2, _p(xy), xy.stride(0), float(center_scale), _p(cnt), cap, _p(out), _stream())
    _lib.check(rc, "cofi_extract_patches")
    return out


def gather_rows_sel(x, row_idx, cnt, cap: int):
    lib = _lib.load()
    _mat(x, "x")
    out = torch.empty((cap, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(lib.cofi_gather_rows_sel(_p(x), _ld(x), x.shape[1], _p(row_idx), _p(cnt), cap, _p(out), _ld(out), _stream()),
               "cofi_gather_rows_sel")
    return out


def fine_match(patches, pc_feats, xy, cnt, center_scale: float):
    """patches (cap,C,16), pc_feats (cap,C), xy (2,ld) -> fine_xy (2,cap), best (cap,) int32."""
    lib = _lib.load()
    cap, C, _ = patches.shape
    fine_xy = torch.empty((2, cap), dtype=torch.float32, device=patches.device)
    best = torch.empty((cap,), dtype=torch.int32, device=patches.device)
    rc = lib.cofi_fine_match(_p(patches), _p(pc_feats), _ld(pc_feats), C, _p(xy), xy.stride(0), float(center_scale), _p(cnt), cap,
                             _p(fine_xy), _p(best), _stream())
    _lib.check(rc, "cofi_fine_match")
    return fine_xy, best
