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
import threading

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_LEAKY01 = 0, 1, 2, 3
GEMM_BF16X3 = 0x100
GEMM_BF16X6 = 0x800
GEMM_L2NORM = 0x1000   # rows L2-normalised in the epilogue (N <= 128)
GEMM_W_F16PRE = 0x4000   # with GEMM_F16X3, launches of its kernel only: W pre-split into fp16 hi / lo with panel scales (pack_f16x3_weight; include/cofi_hip.h)
GEMM_F16X3 = 0x2000    # with GEMM_BF16X6: the large contractions (256 x 128 kernel) in the three-product fp16 split (include/cofi_hip.h)
# arithmetic of the dense contractions: "bf16x3" (default) = 3-term bf16 split (hi*hi + hi*lo + lo*hi) on the bf16 matrix cores
# with fp32 accumulation, ~2^-16 relative error per product; "f32" = exact fp32 MFMA (COFI_GEMM=f32)
GEMM_MODE = os.environ.get("COFI_GEMM", "bf16x3")


# Which intra-frame fork/join branches are taken (see Branch).  With >= 2 frames in flight the frames themselves fill
# the GPU and intra-frame forks only add join overhead; with one frame in flight they shorten the critical path.
BRANCH_MASK = 7


_TLS = threading.local()


def gemm_mode() -> str:
    """The contraction arithmetic of the CALLING thread: the innermost `arithmetic(...)` context this thread is inside, else the process
    default GEMM_MODE (COFI_GEMM).  Everything that depends on the arithmetic - kernel flags, operand planes, hipGraph cache keys - reads
    it through this function."""
    return getattr(_TLS, "mode", None) or GEMM_MODE


# "bf16x6" (the fp32-grade arithmetic): the LARGE contractions - the shapes of the 256 x 128 one-workgroup-per-CU kernel, which sit at the
# chip's power wall with six bf16 products - run in the three-product fp16 split instead (COFI_GEMM_F16X3, csrc/gemm_f16_big.inc: fp16 hi +
# lo of x * 2^e with in-kernel range tracking; same or smaller error against fp64, half the matrix work).  COFI_F16X3=0 switches it off.
F16X3_BIG = os.environ.get("COFI_F16X3", "1") != "0"


# ... and those launches read the STATIC operand (SplitW weights) pre-split: the kernel copies W's two fp16 planes instead of splitting the
# 128 x 32 W tile again in every workgroup (a third of the split work of its K-tile).  COFI_F16X3_WPRE=0: W is split on the fly (A/B).
F16X3_WPRE = os.environ.get("COFI_F16X3_WPRE", "1") != "0"


def pack_f16x3_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 (N, K) weight, K % 4 == 0 -> the COFI_GEMM_W_F16PRE operand (include/cofi_hip.h): a flat fp32-typed buffer of N * K + ceil(N / 128)
    words - the matrix with every aligned group of four values replaced by {hi0..hi3, lo0..lo3} fp16 of w * s (hi = f16(w s) round to nearest
    even, lo = f16(w s - hi)), s = the power of two of the row's 128-row panel that puts the panel's largest |w| into [2^11, 2^12) (the choice
    of f16_scale_for in csrc/gemm_f16_big.inc), followed by the panel scales."""
    w = w.detach()   # (a Parameter of a caller that runs the forward outside torch.no_grad(): the packed form is data, not a graph node)
    N, K = w.shape
    if K % 4:
        raise _lib.CofiError("pack_f16x3_weight: K must be a multiple of 4")
    npan = (N + 127) // 128
    wp = torch.zeros((npan * 128, K), dtype=torch.float32, device=w.device)
    wp[:N] = w
    m = wp.view(npan, 128 * K).abs().amax(1)
    e = (m.view(torch.int32) >> 23) & 0xff                       # biased exponent of the panel maximum
    se = (127 + 11 - (e - 127)).clamp(27, 227)
    scale = torch.where((e == 0) | (e == 255), torch.ones_like(m), (se << 23).view(torch.float32))
    x = wp * scale.repeat_interleave(128)[:, None]               # exact: a power of two
    hi = x.to(torch.float16)
    lo = (x - hi.to(torch.float32)).to(torch.float16)
    packed = torch.cat([hi.view(npan * 128, K // 4, 4), lo.view(npan * 128, K // 4, 4)], dim=2).contiguous().view(npan * 128, 2 * K).view(torch.float32)
    return torch.cat([packed[:N].reshape(-1), scale]).contiguous()


def f16x3_big() -> bool:
    """True if launches of the calling thread's arithmetic may take the f16x3 kernel (part of every hipGraph cache key)."""
    return F16X3_BIG and gemm_mode() == "bf16x6"


def _gemm_flag() -> int:
    m = gemm_mode()
    if m == "bf16x6":
        return GEMM_BF16X6 | (GEMM_F16X3 if F16X3_BIG else 0)
    return GEMM_BF16X3 if m == "bf16x3" else 0


class arithmetic:
    """Context manager: run the enclosed launches with the given contraction arithmetic ("bf16x3" | "bf16x6" | "f32"; None = leave the
    process default, COFI_GEMM).  "bf16x6": three bf16 planes per operand, six products - fp32-grade results on the bf16 matrix cores
    (include/cofi_hip.h COFI_GEMM_BF16X6).  `CoFiI2P(opt, arithmetic=...)` wraps its forwards in it, so models with different arithmetic
    coexist.  The override is THREAD-LOCAL (`gemm_mode()`): a loader thread running inference while another thread trains, or two
    threads driving models of different arithmetic, cannot flip each other's mode; backward passes run on autograd's own thread and
    re-enter the arithmetic their forward node recorded (autograd._Linear)."""

    def __init__(self, mode):
        if mode not in (None, "bf16x3", "bf16x6", "f32"):
            raise ValueError("arithmetic must be 'bf16x3', 'bf16x6' or 'f32', got %r" % (mode,))
        self.mode = mode

    def __enter__(self):
        self.saved = getattr(_TLS, "mode", None)
        if self.mode is not None:
            _TLS.mode = self.mode
        return self

    def __exit__(self, *exc):
        _TLS.mode = self.saved
        return False


GEMM_W_SPLIT = 0x200
# frames per submission from which pending normalisations are finalized ONCE (cofi_norm_finalize -> per-channel scale | shift) instead of
# folded from the statistics partials by every consumer workgroup; 0 = never (COFI_NORM_FINALIZE_FRAMES).  Round 4, same box, bf16x6,
# frames/s without / with: batch 1 479.5 / 483.6, batch 2 515.6 / 524.8, batch 4 567.6 / 581.1, batch 16 620 / 635 (the fold of a
# 320-slab table in each of the thousands of workgroups of a stack-mode launch is real work; rounds 2-3 measured "equal" at batch 1 when the
# chain was a third longer).  Same fixed-order fp64 fold in both forms: identical bits.
FINALIZE_MIN_FRAMES = int(os.environ.get("COFI_NORM_FINALIZE_FRAMES", "1"))
# bf16x6 reading the weights as three pre-split planes (COFI_GEMM_W_SPLIT with 3 planes): bit-identical to splitting W on the fly, and
# SLOWER on MI355X - 6 instead of 4 bytes per weight element through L2 -> LDS cost more than the conversion instructions they replace
# (round 4, same box: 40960 x 1024 x 3072 1333 -> 1410 us; batch-16 pipeline 622 -> 608 frames/s, batch 1 479 = 479).  Off by default.
X6_W_SPLIT = os.environ.get("COFI_X6_W_SPLIT", "0") == "1"


class SplitW:
    """A static GEMM operand (weight) with its bf16 planes, split ONCE (cofi_split_bf16_planes): `planes` (2, N, ldp) = hi / lo for the
    3-term arithmetic, `planes3` (3, N, ldp) = hi / mid / lo for the 6-term one.  Accepted wherever a weight matrix is: the bf16-split
    kernels read the planes of their arithmetic (no on-the-fly conversion of W), the exact-fp32 kernels read `.w`."""

    def __init__(self, w: torch.Tensor):
        lib = _lib.load()
        _mat(w, "w")
        self.w = w
        self.shape, self.device, self.dtype = w.shape, w.device, w.dtype
        N, K = w.shape
        self.ldp = (K + 7) // 8 * 8
        self.planes = torch.empty((2, N, self.ldp), dtype=torch.int16, device=w.device)
        _lib.check(lib.cofi_split_bf16_planes(_p(w), _ld(w), N, K, _p(self.planes), self.ldp, 2, _stream()), "cofi_split_bf16_planes")
        self._planes3 = None
        self._f16pre = None

    @property
    def f16pre(self):
        """the weight as the f16x3 kernel's pre-split operand (pack_f16x3_weight), built on first use (outside graph capture: the packed
        model is warmed eagerly once before any capture)"""
        if self._f16pre is None:
            self._f16pre = pack_f16x3_weight(self.w if self.w.is_contiguous() else self.w.contiguous())
        return self._f16pre

    @property
    def planes3(self):
        """(3, N, ldp) hi / mid / lo planes, built on first use (only the opt-in X6_W_SPLIT path reads them; not inside a graph capture)"""
        if self._planes3 is None:
            N, K = self.w.shape
            self._planes3 = torch.empty((3, N, self.ldp), dtype=torch.int16, device=self.w.device)
            _lib.check(_lib.load().cofi_split_bf16_planes(_p(self.w), _ld(self.w), N, K, _p(self._planes3), self.ldp, 3, _stream()), "cofi_split_bf16_planes")
        return self._planes3

    def numel(self):
        return self.w.numel()

    def dim(self):
        return 2


def presplit(w):
    return w if isinstance(w, SplitW) else SplitW(w)


def _wargs(w, f16_eligible=None):
    """(pointer, leading dimension, extra flag) of a weight operand for the current GEMM mode.  f16_eligible: callable -> bool, asked only for
    a SplitW in the f16x3 configuration: does this launch run on the f16x3 kernel (cofi_gemm_f16x3_eligible / cofi_conv2d_f16x3_eligible)?"""
    if isinstance(w, SplitW):
        if gemm_mode() == "bf16x3":
            return _p(w.planes), w.ldp, GEMM_W_SPLIT
        if gemm_mode() == "bf16x6" and X6_W_SPLIT:
            return _p(w.planes3), w.ldp, GEMM_W_SPLIT
        if f16_eligible is not None and F16X3_WPRE and f16x3_big() and w.shape[1] % 4 == 0 and f16_eligible():
            return _p(w.f16pre), w.shape[1], GEMM_W_F16PRE
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


# Statistics partials -> per-channel scale / shift: finalized once per (statistics, affine pair) by cofi_norm_finalize and read by every
# consumer (FINALIZE_MIN_FRAMES above; round 4), or - COFI_NORM_FINALIZE_FRAMES=0, the form of rounds 2-3 - folded by every consumer
# workgroup itself (stat_fold.h).


# ------------------------------------------------------------------------------------------ dense
class ColStats:
    """Statistics partials a GEMM / convolution epilogue left behind for its (M, C) output: part (nslab, C // width, 2)
    = {sum, sum of squares} per 64-row slab and per `width` adjacent columns (include/cofi_hip.h: cofi_norm_desc_t).
    Consumers fold them in-kernel; `finalize()` (separate launch, width 1 only) is the fallback for layouts the in-kernel
    fold does not take."""

    def __init__(self, part: torch.Tensor, M: int, groups: int, frames: int = 1, eps: float = 1e-5, width: int = 1, slab_rows: int = 64):
        self.part, self.M, self.groups, self.frames, self.eps, self.width = part, M, groups, frames, eps, width
        self.slab_rows = slab_rows   # rows per slab of the producer (64: GEMM / convolution epilogues; kpconv_fused: 64 / 32 / 16)

    @property
    def C(self) -> int:
        return self.part.shape[1] * self.width

    def fusable(self) -> bool:
        C, tc, G = self.C, self.part.shape[1], self.groups
        cpg = C // G
        rows = self.M // self.frames
        return ((tc & (tc - 1)) == 0 and tc >= 2 and (G & (G - 1)) == 0 and cpg % self.width == 0 and G <= 1024
                and (self.frames == 1 or rows % self.slab_rows == 0)
                and self.part.shape[0] == self.frames * ((rows + self.slab_rows - 1) // self.slab_rows))

    def finalize(self) -> torch.Tensor:
        if self.width != 1 or self.slab_rows != 64:
            raise _lib.CofiError("ColStats.finalize needs per-column partials (width 1) in 64-row slabs")
        return group_stats_from_colpart(self.part, self.M, self.groups, self.eps, self.frames)

    def desc(self, gamma=None, beta=None, slope: float = 1.0) -> "_lib.NormDesc":
        d = _lib.NormDesc()
        d.partials = self.part.data_ptr()
        d.nslab, d.width, d.channels, d.groups = self.part.shape[0], self.width, self.C, self.groups
        d.gamma = None if gamma is None else gamma.data_ptr()
        d.beta = None if beta is None else beta.data_ptr()
        d.eps, d.slope = self.eps, slope
        d.slab_rows = self.slab_rows
        d.scale_shift = None
        # cofi_norm_finalize serves groups of <= 64 table columns (norm.hip); a wider group keeps d.scale_shift = None and its consumers
        # fold the partials themselves, as before the finalize launch existed
        if 0 < FINALIZE_MIN_FRAMES <= self.frames and self.part.shape[1] // max(1, min(self.groups, self.part.shape[1])) <= 64:
            # stack-mode batches: one finalize launch per (statistics, affine pair) instead of a fold of the whole table in EVERY consumer
            # workgroup (thousands per launch at batch 16); the tensor is kept on the statistics object, the descriptor holds its address
            d.scale_shift = self.scale_shift(d, gamma, beta).data_ptr()
        return d

    def scale_shift(self, d, gamma, beta) -> torch.Tensor:
        """(frames, 2, C) finalized scale | shift for this affine pair, computed once (cofi_norm_finalize) and kept with the statistics."""
        key = (None if gamma is None else gamma.data_ptr(), None if beta is None else beta.data_ptr())
        cache = self.__dict__.setdefault("_scsh", {})
        t = cache.get(key)
        if t is None:
            lib = _lib.load()
            t = torch.empty((self.frames, 2, self.C), dtype=torch.float32, device=self.part.device)
            _lib.check(lib.cofi_norm_finalize(ctypes.byref(d), self.M, self.frames, _p(t), _stream()), "cofi_norm_finalize")
            cache[key] = t
        return t


class Normed:
    """A raw layer output `y` whose GroupNorm / InstanceNorm (+ affine + LeakyReLU) is still pending:
         value = leaky(gn(y; stats) * gamma + beta, slope).
    A GEMM / convolution that consumes it applies the normalisation in its operand loader (cofi_gemm_f32_fused); anything
    else calls `materialize()` (the stand-alone apply kernel)."""
    MAX_FUSED_CHANNELS = 512

    def __init__(self, y: torch.Tensor, stats: ColStats, gamma=None, beta=None, slope: float = 1.0):
        self.y, self.stats, self.gamma, self.beta, self.slope = y, stats, gamma, beta, slope
        self.shape, self.device, self.dtype = y.shape, y.device, y.dtype

    def numel(self):
        return self.y.numel()

    def fusable(self, tile_rows_ok: bool = True) -> bool:
        return (gemm_mode() in ("bf16x3", "bf16x6") and self.stats.fusable() and self.y.shape[1] <= self.MAX_FUSED_CHANNELS and 0.0 <= self.slope <= 1.0
                and tile_rows_ok)

    def desc(self):
        return self.stats.desc(self.gamma, self.beta, self.slope)

    def materialize(self, out=None, want_row_pos: bool = False):
        return group_norm_apply(self.y, self.stats, self.gamma, self.beta, slope=self.slope, out=out, frames=self.stats.frames,
                                want_row_pos=want_row_pos)


def _a_operand(a, M_rows: int, frames: int):
    """-> (raw matrix, NormDesc or None) for a GEMM A operand that may carry a pending normalisation."""
    if not isinstance(a, Normed):
        return a, None
    # a tile of GEMM rows must lie inside one frame: 128-row tiles are the largest
    if a.fusable(frames == 1 or (M_rows // frames) % 128 == 0):
        return a.y, a.desc()
    return a.materialize(), None


GEMM_A_SPLIT = 0x400


class SplitA:
    """An activation that its producer already wrote as bf16 hi / lo planes (kpconv_aggregate(planes=True)): planes (2, M, ld) int16,
    logical shape (M, K).  Only the bf16x3 GEMM with a pre-split weight consumes it (COFI_GEMM_A_SPLIT)."""

    def __init__(self, planes: torch.Tensor, K: int):
        self.planes, self.K = planes, K
        self.shape, self.device, self.dtype = (planes.shape[1], K), planes.device, torch.float32


def _gemm_impl(a, w, out, bias, rowdiv, act, stat_width, frames, l2norm=False):
    lib = _lib.load()
    if isinstance(a, Normed) and not isinstance(w, SplitW) and a.fusable():
        w = SplitW(w)   # the normalising loader lives in the pre-split-weight kernels
    M0 = a.shape[0]
    aflag = 0
    if isinstance(a, SplitA):
        if gemm_mode() != "bf16x3" or not isinstance(w, SplitW):
            raise _lib.CofiError("gemm: a pre-split activation needs the bf16x3 arithmetic and a pre-split weight")
        asplit, nd, aflag = a, None, GEMM_A_SPLIT
        a_ptr, a_ld, (M, K) = _p(a.planes), a.planes.shape[2], a.shape
    else:
        a, nd = _a_operand(a, M0, frames)
        _mat(a, "a")
        a_ptr, a_ld, (M, K) = _p(a), _ld(a), a.shape
    if not isinstance(w, SplitW):
        _mat(w, "w")
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.CofiError("gemm: K mismatch %s vs %s" % (tuple(a.shape), tuple(w.shape)))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _mat(out, "out")
    _vec(bias, "bias", N), _vec(rowdiv, "rowdiv", M)
    colpart = None
    if stat_width:
        if N % stat_width:
            raise _lib.CofiError("gemm: %d columns are not a multiple of the statistics width %d" % (N, stat_width))
        colpart = torch.empty((lib.cofi_gemm_f32_stat_slabs(M, N, K), N // stat_width, 2), dtype=torch.float32, device=a.device)
    if M == 0:   # zero rows: nothing to launch (torch hands out a null pointer for an empty tensor, which the C ABI rejects)
        return out, colpart
    ws = _WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, N, K), a.device)
    wp, wld, wflag = _wargs(w, None if (l2norm or aflag) else (lambda: lib.cofi_gemm_f16x3_eligible(M, N, K, int(nd is not None), frames) == 1))
    if l2norm and N > 128:
        raise _lib.CofiError("gemm: the L2-normalising epilogue serves N <= 128")
    rc = lib.cofi_gemm_f32_fused(a_ptr, a_ld, None if nd is None else ctypes.byref(nd), wp, wld, _p(out), _ld(out), M, N, K, _p(bias),
                                 _p(rowdiv), act | _gemm_flag() | wflag | aflag | (GEMM_L2NORM if l2norm else 0), _p(colpart), max(stat_width, 1), _p(ws), 0 if ws is None else ws.numel(),
                                 frames, _stream())
    _lib.check(rc, "cofi_gemm_f32_fused")
    return out, colpart


def gemm(a, w, out: Optional[torch.Tensor] = None, bias=None, rowdiv=None, act: int = ACT_NONE, frames: int = 1, l2norm: bool = False):
    """out[m,n] = act( (a @ w.T)[m,n] / rowdiv[m] + bias[n] );  a (M,K) [or a Normed: normalised on the fly], w (N,K).
    l2norm (N <= 128): the rows of the result are L2-normalised in the epilogue (F.normalize(dim=1))."""
    return _gemm_impl(a, w, out, bias, rowdiv, act, 0, frames, l2norm)[0]


def gemm_colstats(a, w, out=None, bias=None, rowdiv=None, act: int = ACT_NONE, stat_width: int = 1, frames: int = 1):
    """gemm() that also returns the fused statistics partials: (out, colpart (nslab, N // stat_width, 2))."""
    return _gemm_impl(a, w, out, bias, rowdiv, act, stat_width, frames)


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


def col_inv_norm_from_colpart(colpart, M: int, C: int, eps: float = 1e-12, frames: int = 1):
    """M = rows of the activation (all frames) the per-column partials were taken over."""
    lib = _lib.load()
    nslab, ncols, _ = colpart.shape
    out = torch.empty((C,) if frames == 1 else (frames, C), dtype=torch.float32, device=colpart.device)
    _lib.check(lib.cofi_col_inv_norm_from_colpart(_p(colpart), nslab, M, ncols, C, eps, _p(out), frames, _stream()),
               "cofi_col_inv_norm_from_colpart")
    return out


# ------------------------------------------------------------------------------------------ KPConv
def row_sum_positive(feats: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _mat(feats, "feats")
    out = torch.empty((feats.shape[0],), dtype=torch.uint8, device=feats.device)
    _lib.check(lib.cofi_row_sum_positive(_p(feats), _ld(feats), feats.shape[0], feats.shape[1], _p(out), _stream()), "cofi_row_sum_positive")
    return out


def kpconv_aggregate(feats, q_pts, s_pts, idx, kernel_points, sigma: float, row_pos=None, frames: int = 1, order=None, planes: bool = False):
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
    planes = planes and C % 4 == 0 and C > 4 and (15 * C) % 8 == 0
    if planes:   # bf16 hi / lo planes: the part-2 GEMM takes them without any conversion (SplitA)
        agg = torch.empty((2, M, 15 * C), dtype=torch.int16, device=feats.device)
    else:
        agg = torch.empty((M, 15 * C), dtype=torch.float32, device=feats.device)
    cnt = torch.empty((M,), dtype=torch.float32, device=feats.device)
    if C <= 4 and row_pos is None and M > 0:
        # first layer: [features | position | positive-sum flag] packed into 32-byte records, one gather per neighbour
        recs = torch.empty((N, 8), dtype=torch.float32, device=feats.device)
        _lib.check(lib.cofi_kp_pack_c4(_p(feats), _ld(feats), C, _p(s_pts), N, _p(recs), _stream()), "cofi_kp_pack_c4")
        rc = lib.cofi_kpconv_aggregate_c4(_p(recs), N // frames, C, _p(q_pts), _p(idx), M // frames, H, _p(kernel_points), float(sigma), _p(agg),
                                          15 * C, _p(cnt), frames, _p(order), _stream())
        _lib.check(rc, "cofi_kpconv_aggregate_c4")
        return agg, cnt
    if row_pos is None:
        row_pos = row_sum_positive(feats)
    rc = lib.cofi_kpconv_aggregate(_p(feats), _ld(feats), N // frames, C, _p(q_pts), _p(s_pts), _p(idx), M // frames, H,
                                   _p(kernel_points), float(sigma), _p(row_pos), _p(agg), 15 * C, int(planes), _p(cnt), frames, _p(order), _stream())
    _lib.check(rc, "cofi_kpconv_aggregate")
    return (SplitA(agg, 15 * C) if planes else agg), cnt


def kpconv_fused_slab_rows(C: int, M: int, frames: int = 1) -> int:
    """Rows per statistics slab of `kpconv_fused` for M queries per frame; 0 = shape not served by the fused kernel."""
    if gemm_mode() != "bf16x3":
        return 0
    return int(_lib.load().cofi_kpconv_fused_slab_rows(C, M, frames))


def kpconv_fused(feats, q_pts, s_pts, idx, kernel_points, sigma: float, w: "SplitW", bias, stat_width: int = 1, row_pos=None, frames: int = 1,
                 order=None):
    """The whole KPConv operator (kpconv.py:91-116) of a narrow layer (C = 32 / 64 in = out channels) in one launch: -> (y (M, C),
    statistics partials (M / slab_rows, C / stat_width, 2), slab_rows).  `w`: the (C, 15 C) packed weight as pre-split planes."""
    lib = _lib.load()
    _mat(feats, "feats"), _mat(idx, "idx", torch.int32)
    if not (q_pts.is_contiguous() and s_pts.is_contiguous() and kernel_points.is_contiguous() and idx.is_contiguous()):
        raise _lib.CofiError("kpconv_fused: points / idx / kernel_points must be contiguous")
    N, C = feats.shape
    M, H = idx.shape
    if s_pts.shape != (N, 3) or q_pts.shape != (M, 3) or kernel_points.shape != (15, 3) or N % frames or M % frames:
        raise _lib.CofiError("kpconv_fused: shape mismatch")
    if not isinstance(w, SplitW) or tuple(w.shape) != (C, 15 * C):
        raise _lib.CofiError("kpconv_fused: the weight must be the pre-split (C, 15 C) packing")
    sr = kpconv_fused_slab_rows(C, M // frames, frames)
    if sr == 0:
        raise _lib.CofiError("kpconv_fused: shape not supported (C in {32, 64}, M % 16 == 0, bf16x3 arithmetic)")
    if row_pos is None:
        row_pos = getattr(feats, "cofi_row_pos", None)   # left there by the group_norm_apply that produced feats
    if row_pos is None:
        row_pos = row_sum_positive(feats)
    y = torch.empty((M, C), dtype=torch.float32, device=feats.device)
    part = torch.empty((M // sr, C // stat_width, 2), dtype=torch.float32, device=feats.device)
    rc = lib.cofi_kpconv_fused(_p(feats), _ld(feats), N // frames, C, _p(q_pts), _p(s_pts), _p(idx), M // frames, H, _p(kernel_points), float(sigma),
                               _p(row_pos), _p(w.planes), w.ldp, _p(bias), _p(y), _ld(y), _p(part), stat_width, frames, _p(order), _stream())
    _lib.check(rc, "cofi_kpconv_fused")
    return y, part, sr


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
def group_stats(x, groups: int, eps: float = 1e-5, frames: int = 1, exact: bool = False):
    """-> stats (groups,2), or (frames, groups, 2) in stack mode (x = frames blocks of M/frames rows).  exact: every sum in fp64
    (cofi_group_stats_exact, the training path)."""
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    stats = torch.empty((groups, 2) if frames == 1 else (frames, groups, 2), dtype=torch.float32, device=x.device)
    ws = _WS_STATS.get(lib.cofi_group_stats_workspace(M, C, groups, frames), x.device)
    fn = lib.cofi_group_stats_exact if exact else lib.cofi_group_stats
    _lib.check(fn(_p(x), _ld(x), M, C, groups, eps, _p(stats), _p(ws), ws.numel(), frames, _stream()), "cofi_group_stats")
    return stats


def group_norm_apply(x, stats, gamma=None, beta=None, slope: float = 1.0, res=None, res_stats=None, res_gamma=None, res_beta=None,
                     out=None, frames: int = 1, want_row_pos: bool = False):
    """stats: (groups,2) / stack mode (frames, groups, 2) tensor, or ColStats (statistics partials, folded in-kernel).
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
        nd = stats.desc(gamma, beta, slope)
        rnd = None if res_stats is None else res_stats.desc(res_gamma, res_beta, 1.0)
        rc = lib.cofi_group_norm_apply_partials(_p(x), _ld(x), M, C, ctypes.byref(nd), _p(res), 0 if res is None else _ld(res),
                                                None if rnd is None else ctypes.byref(rnd), _p(out), _ld(out), _p(row_pos), frames, _stream())
        _lib.check(rc, "cofi_group_norm_apply_partials")
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


def layer_norm_act(x, gamma, beta, slope: float = 1.0, res=None, res_first: bool = True, out=None, eps: float = 1e-5):
    """y = leaky(LN(x) * gamma + beta + (res if res_first), slope) + (res if not res_first)   (cofi_layer_norm_act)."""
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    rc = lib.cofi_layer_norm_act(_p(x), _ld(x), M, C, _p(gamma), _p(beta), eps, float(slope), _p(res), 0 if res is None else _ld(res),
                                 int(res_first), _p(out), _ld(out), _stream())
    _lib.check(rc, "cofi_layer_norm_act")
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


def l2norm_rows2(x, out, out2):
    """F.normalize(x, dim=1) written to two row-major destinations in one launch."""
    lib = _lib.load()
    _mat(x, "x"), _mat(out, "out"), _mat(out2, "out2")
    M, C = x.shape
    if M == 0:
        return out, out2
    _lib.check(lib.cofi_l2norm_rows2(_p(x), _ld(x), M, C, _p(out), _ld(out), _p(out2), _ld(out2), _stream()), "cofi_l2norm_rows2")
    return out, out2


def col_mean(x, frames: int = 1):
    """(frames * rows, C) -> (frames, C) column means per frame (AdaptiveAvgPool2d(1) of an NHWC map)."""
    lib = _lib.load()
    _mat(x, "x")
    out = torch.empty((frames, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(lib.cofi_col_mean(_p(x), _ld(x), x.shape[0], x.shape[1], _p(out), frames, _stream()), "cofi_col_mean")
    return out


def transpose(x, out=None, frames: int = 1):
    """(M, C) -> (C, M); frames > 1: x holds `frames` row blocks of M / frames rows, out = (frames, C, M / frames), one launch."""
    lib = _lib.load()
    _mat(x, "x")
    M, C = x.shape
    if M % frames:
        raise _lib.CofiError("transpose: rows are not a multiple of the frame count")
    M //= frames
    if out is None:
        out = torch.empty((C, M) if frames == 1 else (frames, C, M), dtype=torch.float32, device=x.device)
    _lib.check(lib.cofi_transpose(_p(x), _ld(x), M, C, _p(out), out.stride(-2), frames, _stream()), "cofi_transpose")
    return out


def transpose_pair(a, b):
    """(a^T, b^T) of two matrices with the same number of rows, one launch."""
    lib = _lib.load()
    _mat(a, "a"), _mat(b, "b")
    M = a.shape[0]
    if b.shape[0] != M:
        raise _lib.CofiError("transpose_pair: both matrices must have the same number of rows")
    at = torch.empty((a.shape[1], M), dtype=torch.float32, device=a.device)
    bt = torch.empty((b.shape[1], M), dtype=torch.float32, device=a.device)
    _lib.check(lib.cofi_transpose_pair(_p(a), _ld(a), a.shape[1], _p(at), _ld(at), _p(b), _ld(b), b.shape[1], _p(bt), _ld(bt), M, _stream()),
               "cofi_transpose_pair")
    return at, bt


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
                colstats: bool = False, out=None, frames: int = 1, stat_width: int = 1, act_col0: int = 0, l2norm: bool = False):
    """Implicit-GEMM convolution on an NHWC map x (H*W, Cin) [row-major view, any leading dimension; or a Normed: the pending
    InstanceNorm + ReLU of the previous convolution is applied by the operand loader]; w (Cout, ks*ks*Cin).
    act applies to output columns >= act_col0 (stacked filters of two convolutions of the same input).
    -> y (Ho*Wo, Cout) [, colpart]."""
    lib = _lib.load()
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    nd = None
    if isinstance(x, Normed):
        if isinstance(w, SplitW) and x.fusable(frames == 1 or (Ho * Wo) % 128 == 0):
            x, nd = x.y, x.desc()
        else:
            x = x.materialize()
    _mat(x, "x")
    if not isinstance(w, SplitW):
        _mat(w, "w")
    Cin, Cout = x.shape[1], w.shape[0]
    if x.shape[0] != frames * H * W or w.shape[1] != ks * ks * Cin:
        raise _lib.CofiError("conv2d_nhwc: shape mismatch x %s w %s H %d W %d ks %d" % (tuple(x.shape), tuple(w.shape), H, W, ks))
    M, K = frames * Ho * Wo, ks * ks * Cin
    if out is None:
        out = torch.empty((M, Cout), dtype=torch.float32, device=x.device)
    part = None
    if colstats:
        part = torch.empty((lib.cofi_gemm_f32_stat_slabs(M, Cout, K), Cout // stat_width, 2), dtype=torch.float32, device=x.device)
    ws = _WS_GEMM.get(lib.cofi_gemm_f32_workspace(M, Cout, K), x.device)
    # convolution weights are dense (Cout, K) / planes (2, Cout, roundup8(K)) / the f16x3 kernel's pre-split form
    wp, _wld, wflag = _wargs(w, None if l2norm else (lambda: lib.cofi_conv2d_f16x3_eligible(H, W, Cin, Cout, ks, stride, pad, _ld(x), int(nd is not None), frames) == 1))
    rc = lib.cofi_conv2d_nhwc_fused(_p(x), _ld(x), None if nd is None else ctypes.byref(nd), H, W, Cin, wp, Cout, ks, stride, pad, _p(bias),
                                    _p(res), 0 if res is None else _ld(res), act | _gemm_flag() | wflag | (GEMM_L2NORM if l2norm else 0), act_col0, _p(out), _ld(out), _p(part), stat_width,
                                    _p(ws), 0 if ws is None else ws.numel(), frames, _stream())
    _lib.check(rc, "cofi_conv2d_nhwc_fused")
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


# ------------------------------------------------------------------------------------------ attention
_WS_ATTN = Workspace()
_WS_ATTN_KV = Workspace()
# bf16x6 attention: K / V are cut into their bf16 planes ONCE per call (cofi_attention_kv_planes, one extra launch) when a frame has at least
# this many QUERY rows - every 64 of them are a workgroup that otherwise repeats the split of the frame's whole K / V.  Measured on MI355X
# (tools/attn_presplit_probe.py, profiles/r06/attn_presplit_probe.txt; us per call, split in every workgroup -> split launch + kernel):
# L = 1280 (KITTI; 16 frames) 113.8 -> 11.5 + 100.2, a tie, and a loss on one frame (12.7 -> 3.5 + 12.2); L = 2560 224 -> 12 + 189;
# L = 22400 (stress) 243 -> 4 + 199 and 1580 -> 12 + 1462.  Hence the threshold.  COFI_ATTN_PRESPLIT_ROWS=0: never, 1: always.
ATTN_PRESPLIT_ROWS = int(os.environ.get("COFI_ATTN_PRESPLIT_ROWS", "2048"))


class AttnParts:
    """Output of the attention kernel in its native form: per (frame, head, 32-query block) partial slots
    (csrc/attention_parts.h).  `ops.loftr_tail` merges them in its loader; `merge()` writes the plain (frames*L, H*D) matrix."""

    def __init__(self, buf, L: int, S: int, H: int, D: int, frames: int):
        self.buf, self.L, self.S, self.H, self.D, self.frames = buf, L, S, H, D, frames
        self.shape = (frames * L, H * D)

    def merge(self, out=None):
        lib = _lib.load()
        if out is None:
            out = torch.empty(self.shape, dtype=torch.float32, device=self.buf.device)
        _mat(out, "out")
        _lib.check(lib.cofi_attention_merge(_p(self.buf), self.buf.numel(), self.L, self.S, self.H, self.D, self.frames, _p(out), _ld(out),
                                            _stream()), "cofi_attention_merge")
        return out


ATTN_MODE = os.environ.get("COFI_ATTN", "auto")   # "auto" | "bf16x6" | "f32"


def attention_arith() -> str:
    """Arithmetic of the attention kernel: the fp32-grade bf16 split ("bf16x6", 417 TF/s roof) unless the dense contractions run on the
    exact fp32 matrix instruction (gemm_mode() "f32" - the calling thread's `arithmetic(...)` context, else COFI_GEMM: the attention then
    does too, 157 TF/s roof).  COFI_ATTN=f32 / bf16x6 overrides."""
    if ATTN_MODE in ("bf16x6", "f32"):
        return ATTN_MODE
    return "f32" if gemm_mode() == "f32" else "bf16x6"


def attention_parts(q, k, v, q_colscale=None, nhead: int = 4, frames: int = 1, q_colpart=None, q_eps: float = 1e-12) -> "AttnParts":
    """The attention kernel itself (cofi_attention_parts): -> the partial slots (AttnParts) in a per-stream workspace.
    Stack mode: q (frames*L, HD), k/v (frames*S, HD), q_colscale (frames, HD).  q_colpart (nslab, ncols, 2): the column
    partials of the GEMM / fused tail that produced q (its first HD columns) - the token-axis norm of Q is then folded inside the kernel."""
    lib = _lib.load()
    _mat(q, "q"), _mat(k, "k"), _mat(v, "v")
    L, HD = q.shape
    S = k.shape[0]
    D = HD // nhead
    if L % frames or S % frames:
        raise _lib.CofiError("attention: rows are not a multiple of the frame count")
    L, S = L // frames, S // frames
    nbytes = lib.cofi_attention_workspace(L, S, nhead, D, frames)
    if nbytes == 0:
        raise _lib.CofiError("attention: unsupported shape (head dimension must be 32)")
    # the slot table outlives this call when it is handed to the consumer: a per-stream workspace is safe (stream ordered)
    ws = _WS_ATTN.get(nbytes, q.device)
    if attention_arith() == "bf16x6" and 0 < ATTN_PRESPLIT_ROWS <= L:
        pb = lib.cofi_attention_kv_planes_bytes(S, nhead, D, frames)
        img = _WS_ATTN_KV.get(pb, q.device)
        _lib.check(lib.cofi_attention_kv_planes(_p(k), _ld(k), _p(v), _ld(v), S, nhead, D, frames, _p(img), img.numel(), _stream()), "cofi_attention_kv_planes")
        rc = lib.cofi_attention_parts_planes(_p(q), _ld(q), _p(img), img.numel(), _p(q_colscale), _p(q_colpart),
                                             0 if q_colpart is None else q_colpart.shape[0], 0 if q_colpart is None else q_colpart.shape[1], q_eps,
                                             L, S, nhead, D, 1.0 / math.sqrt(D), frames, _p(ws), ws.numel(), _stream())
        _lib.check(rc, "cofi_attention_parts_planes")
        return AttnParts(ws, L, S, nhead, D, frames)
    fn = lib.cofi_attention_parts_bf16x6 if attention_arith() == "bf16x6" else lib.cofi_attention_parts
    rc = fn(_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(q_colscale), _p(q_colpart),
            0 if q_colpart is None else q_colpart.shape[0], 0 if q_colpart is None else q_colpart.shape[1], q_eps,
            L, S, nhead, D, 1.0 / math.sqrt(D), frames, _p(ws), ws.numel(), _stream())
    _lib.check(rc, "cofi_attention_parts")
    return AttnParts(ws, L, S, nhead, D, frames)


def attention(q, k, v, q_colscale=None, nhead: int = 4, out=None, frames: int = 1, q_colpart=None, q_eps: float = 1e-12,
              parts: bool = False):
    """attention_parts + (parts=False) the merge of the partial slots into the plain (frames*L, H*D) matrix; parts=True: the slots
    themselves (the fused layer tail merges them in its loader)."""
    res = attention_parts(q, k, v, q_colscale=q_colscale, nhead=nhead, frames=frames, q_colpart=q_colpart, q_eps=q_eps)
    return res if parts else res.merge(out)


def split_bf16(w: torch.Tensor):
    """fp32 -> (hi, lo) bf16 planes as int16 tensors: hi = bf16(w) (RNE), lo = bf16(w - hi)."""
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    return hi.contiguous().view(torch.int16), lo.contiguous().view(torch.int16)


def split_planes(w: torch.Tensor, n: int) -> torch.Tensor:
    """fp32 (N, K) -> (n, N, K) bf16 planes as int16: plane 0 = bf16(w) (RNE), every further plane = bf16 of what the planes before it
    left over (n = 2: the hi / lo pair of the 3-term split, n = 3: hi / mid / lo = all 24 mantissa bits, the 6-term split)."""
    r = w.to(torch.float32)
    out = []
    for _ in range(n):
        p = r.to(torch.bfloat16)
        out.append(p.view(torch.int16))
        r = r - p.to(torch.float32)
    return torch.stack(out).contiguous()


def fragment_order(planes: torch.Tensor) -> torch.Tensor:
    """(n, N, K) bf16 planes -> the same values in MFMA-FRAGMENT ORDER (cofi_loftr_tail_desc_t::w_frag, include/cofi_hip.h): per plane, row
    block T = n / 32 and k-step s = k / 16 the 64 lanes x 8 bf16 of a wave's B operand, contiguous (1 KB).  Shape kept (n, N, K) - only the
    order in memory changes."""
    n, N, K = planes.shape
    if N % 32 or K % 16:
        raise _lib.CofiError("fragment_order: N must be a multiple of 32 and K of 16")
    return planes.view(n, N // 32, 32, K // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(n, N, K)


# the fused layer tail reads its weight planes in fragment order (one contiguous 1 KB segment per wave load instead of 32 row pieces of 32 B);
# COFI_TAIL_FRAG=0: the row-major planes (A/B; identical bits)
TAIL_FRAG = os.environ.get("COFI_TAIL_FRAG", "1") != "0"


def tail_suffix() -> str:
    """Key suffix of the tail's weight planes in a packed layer (transformer.pack_layer): ".pN" row-major, ".fN" fragment order."""
    return (".f%d" if TAIL_FRAG else ".p%d") % tail_planes()


def tail_planes() -> int:
    """bf16 planes per operand of the fused layer tail for the current arithmetic; 0 = not served (exact fp32)."""
    return {"bf16x3": 2, "bf16x6": 3}.get(gemm_mode(), 0)


def loftr_tail(msg, x, w, out, eps: float = 1e-5, proj=(), out_l2=None, out_l2t=None):
    """out = x + LN2(relu([x | LN1(msg Wm^T)] W0^T) W2^T) in one kernel (cofi_loftr_tail); `w` holds the pre-split planes
    ("merge.p2" / ".p3", ...).  msg: the attention output as a matrix or as AttnParts (merged by the kernel's loader).
    proj: up to two (planes (p, N, 128) int16, y (rows, N) view, part (rows / 32, N, 2) or None) - projections of `out` computed in the
    same launch (the next layers' q / k / v); out_l2 / out_l2t: F.normalize(out, dim=1) token-major / channel-major."""
    lib = _lib.load()
    _mat(x, "x"), _mat(out, "out")
    npl = tail_planes()
    if npl == 0:
        raise _lib.CofiError("loftr_tail serves the bf16x3 / bf16x6 arithmetics")
    sfx = tail_suffix()
    d = _lib.TailDesc()
    d.w_frag = 1 if TAIL_FRAG else 0   # (the projection planes in `proj` must come from the same suffix)
    keep = []   # tensors whose addresses the descriptor holds (alive until the launch is enqueued)
    if isinstance(msg, AttnParts):
        d.parts, d.parts_bytes, d.L, d.S, d.H, d.frames = msg.buf.data_ptr(), msg.buf.numel(), msg.L, msg.S, msg.H, msg.frames
    else:
        _mat(msg, "msg")
        d.msg, d.ldm, d.rows, d.frames = msg.data_ptr(), _ld(msg), msg.shape[0], 1
    d.x, d.ldx, d.planes = x.data_ptr(), _ld(x), npl
    d.wm, d.w0, d.w2 = w["merge" + sfx].data_ptr(), w["mlp.0" + sfx].data_ptr(), w["mlp.2" + sfx].data_ptr()
    d.n1_gamma, d.n1_beta, d.n2_gamma, d.n2_beta = (w[k].data_ptr() for k in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"))
    d.eps, d.out, d.ldo = eps, out.data_ptr(), _ld(out)
    if len(proj) > 2:
        raise _lib.CofiError("loftr_tail: at most two projection segments")
    for i, (pw, y, part) in enumerate(proj):
        N = pw.shape[1]
        if pw.dtype != torch.int16 or pw.shape[0] != npl or pw.shape[2] != 128 or not pw.is_contiguous():
            raise _lib.CofiError("loftr_tail: projection weights must be (%d, N, 128) int16 planes" % npl)
        _mat(y, "proj_y")
        if y.shape[1] != N or y.shape[0] != out.shape[0]:
            raise _lib.CofiError("loftr_tail: projection output must be (rows, %d)" % N)
        if part is not None and (tuple(part.shape) != ((out.shape[0] + 31) // 32, N, 2) or not part.is_contiguous() or part.dtype != torch.float32):
            raise _lib.CofiError("loftr_tail: projection partials must be a contiguous (rows / 32, N, 2) float32 tensor")
        d.proj_n[i], d.proj_w[i], d.proj_y[i], d.proj_ldy[i] = N, pw.data_ptr(), y.data_ptr(), _ld(y)
        d.proj_part[i] = None if part is None else part.data_ptr()
        keep += [pw, y, part]
    if out_l2 is not None:
        _mat(out_l2, "out_l2")
        d.out_l2, d.ld_l2 = out_l2.data_ptr(), _ld(out_l2)
    if out_l2t is not None:
        _mat(out_l2t, "out_l2t")
        d.out_l2t, d.ld_l2t = out_l2t.data_ptr(), _ld(out_l2t)
    _lib.check(lib.cofi_loftr_tail(ctypes.byref(d), _stream()), "cofi_loftr_tail")
    return out


MULTI_COPY_BLOCKS = 64   # workgroups per record (the largest record of a frame is ~10 MB)


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
        _lib.check(lib.cofi_multi_copy(_p(table), len(pairs), MULTI_COPY_BLOCKS, _stream()), "cofi_multi_copy")


# ------------------------------------------------------------------------------------------ KNN / indices
KNN_GRID_MIN_SUPPORT = 1024   # smaller support sets: brute force


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


def knn_up_nearest(points, neighbors, sub, out=None):
    """-> (N, 1) int32: for every stage-i point the nearest stage-(i+1) point = column 0 of `knn(points[sub], points, k)`, derived from the
    point's own neighbour row (cofi_knn_up_nearest) instead of searched.  points (N, 3), neighbors (N, k) int32 into the same points,
    sub (S1,) int32: stage-(i+1) point j is stage-i point sub[j]."""
    lib = _lib.load()
    _check_xyz(points, "points")
    _mat(neighbors, "neighbors", torch.int32)
    if sub.dtype != torch.int32 or not sub.is_cuda or not sub.is_contiguous() or not neighbors.is_contiguous():
        raise _lib.CofiError("knn_up_nearest: contiguous CUDA int32 tables expected")
    N, k = neighbors.shape
    if out is None:
        out = torch.empty((N, 1), dtype=torch.int32, device=points.device)
    scratch = torch.empty((N,), dtype=torch.int32, device=points.device)
    _lib.check(lib.cofi_knn_up_nearest(_p(points), N, _p(neighbors), k, _p(sub), sub.numel(), _p(scratch), _p(out), out.stride(0), _stream()),
               "cofi_knn_up_nearest")
    return out


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


def select_matches(score, pix, W8: int, H8: int, thresholds: np.ndarray, min_matches: int = 4, x_max: int = 62, y_max: int = 18, frames: int = 1):
    """-> sel (N,) int32, coarse_xy (2,N) float32, count_dev (2,) int32 [n, threshold index] - with frames > 1 (score / pix hold `frames` blocks
    of N points) (frames, N), (frames, 2, N), (frames, 2) from one launch.  Border rule of the reference
    (model/network.py:184): 2 <= x <= 62, 2 <= y <= 18 - hard-coded there for the KITTI 64x20 map and kept for every image size."""
    lib = _lib.load()
    if score.numel() % frames or not score.is_contiguous() or not pix.is_contiguous():
        raise _lib.CofiError("select_matches: contiguous score / pix of frames * N elements expected")
    N = score.numel() // frames
    lead = () if frames == 1 else (frames,)
    sel = torch.empty(lead + (N,), dtype=torch.int32, device=score.device)
    xy = torch.empty(lead + (2, N), dtype=torch.float32, device=score.device)
    cnt = torch.empty(lead + (2,), dtype=torch.int32, device=score.device)
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    rc = lib.cofi_select_matches(_p(score), _p(pix), N, W8, H8, x_max, y_max, thr.ctypes.data_as(ctypes.c_void_p), len(thr), min_matches,
                                 _p(sel), _p(xy), _p(cnt), frames, _stream())
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


def match_finish(pts4, pts1, sel, cnt, fmap, H2: int, W2: int, xy, fine_pc_all, center_scale: float, frames: int = 1):
    """The tail of a test-mode forward in one launch (cofi_match_finish): -> coarse_pts (cap,3), patches (cap,C,16), fine_pc (cap,C),
    fine_xy (2,cap), best (cap,) - bit-identical to gather_points_sel + nearest_node_sel + gather_rows_sel + extract_patches_nhwc +
    fine_match.  frames > 1 (stack mode: every input holds `frames` equally sized blocks; sel (frames, cap), xy (frames, 2, cap), cnt (frames, 2)):
    the outputs gain a leading frame axis."""
    lib = _lib.load()
    _mat(fmap, "fmap"), _mat(fine_pc_all, "fine_pc_all")
    if sel.numel() % frames or pts4.shape[0] % frames or pts1.shape[0] % frames or fmap.shape[0] != frames * H2 * W2 or not (sel.is_contiguous() and xy.is_contiguous()):
        raise _lib.CofiError("match_finish: shape mismatch")
    cap, C, dev = sel.numel() // frames, fmap.shape[1], fmap.device
    lead = () if frames == 1 else (frames,)
    coarse_pts = torch.empty(lead + (cap, 3), dtype=torch.float32, device=dev)
    patches = torch.empty(lead + (cap, C, 16), dtype=torch.float32, device=dev)
    fine_pc = torch.empty(lead + (cap, C), dtype=torch.float32, device=dev)
    fine_xy = torch.empty(lead + (2, cap), dtype=torch.float32, device=dev)
    best = torch.empty(lead + (cap,), dtype=torch.int32, device=dev)
    rc = lib.cofi_match_finish(_p(pts4.contiguous()), _p(pts1.contiguous()), pts1.shape[0] // frames, _p(sel), _p(cnt), cap, _p(fmap), _ld(fmap), C, H2, W2,
                               _p(xy), xy.stride(-2), float(center_scale), _p(fine_pc_all), _ld(fine_pc_all), _p(coarse_pts), _p(patches), _p(fine_pc),
                               C, _p(fine_xy), _p(best), pts4.shape[0] // frames, frames, _stream())
    _lib.check(rc, "cofi_match_finish")
    return coarse_pts, patches, fine_pc, fine_xy, best
