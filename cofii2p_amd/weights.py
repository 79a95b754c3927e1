"""Deterministic, name-keyed synthetic weights.

There is no network access (no released checkpoint), so both sides of every parity
test — the reference imported in the development container and this package — are
filled from the same generator: ``value(name, shape) = PRNG(seed = crc32(name) ^ salt)``.
Kernel points are *data* in the reference's state_dict (buffers
``*.KPConv.kernel_points``; model/kpconv/kpconv.py:63-65), so a fixed table stands in for
the reference's optimised + randomly rotated disposition
(model/kpconv/kernel_points.py:389-455).
"""
import zlib

import numpy as np


def kernel_point_table(num_kpoints: int, radius: float) -> np.ndarray:
    """15-point rigid kernel disposition: centre + 6 axis + 8 cube-corner directions at
    0.66*radius, turned by a fixed rotation so that no kernel point is axis aligned."""
    assert num_kpoints == 15, "only the K=15 disposition used by CoFiI2P is tabulated"
    dirs = [(0.0, 0.0, 0.0)]
    for ax in range(3):
        for s in (1.0, -1.0):
            v = [0.0, 0.0, 0.0]
            v[ax] = s
            dirs.append(tuple(v))
    for sx in (1.0, -1.0):
        for sy in (1.0, -1.0):
            for sz in (1.0, -1.0):
                dirs.append((sx / 3 ** 0.5, sy / 3 ** 0.5, sz / 3 ** 0.5))
    pts = np.asarray(dirs, dtype=np.float64) * 0.66
    a, b = 0.37, 0.81  # fixed Euler angles
    rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    pts = pts @ (rz @ rx).T
    return (pts * radius).astype(np.float32)


def _rng(name: str, salt: int) -> np.random.Generator:
    return np.random.default_rng((zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1)) & 0xFFFFFFFF)


def synth_tensor(name: str, shape, dtype: str, salt: int = 0) -> np.ndarray:
    """One state_dict entry.  Scales keep activations O(1) through ~60 layers."""
    shape = tuple(shape)
    g = _rng(name, salt)
    leaf = name.rsplit(".", 1)[-1]
    if dtype == "int64":  # BatchNorm num_batches_tracked
        return np.zeros(shape, dtype=np.int64)
    if leaf == "kernel_points":
        raise ValueError("kernel_points come from kernel_point_table")
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "bias":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if name == "pc_score_layer.6.weight":
        # with plain random weights the point score head never reaches the
        # 0.9 acceptance threshold (network.py:147); |w|*0.72 puts ~30 % of the super-points above
        # it on the synthetic frames, the regime of a trained checkpoint (n ~ 300 matches).
        bound = (3.0 / shape[1]) ** 0.5
        return (0.72 * np.abs(g.uniform(-bound, bound, shape))).astype(np.float32)
    if leaf in ("weight", "weights"):
        if len(shape) == 1:  # affine scale of a norm layer
            return (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
        if leaf == "weights":  # KPConv (K, Cin, Cout): contraction over K*Cin, ~1/3 of K active
            fan_in = shape[0] * shape[1] / 3.0
        else:  # Linear (out, in) / ConvNd (out, in, *k)
            fan_in = int(np.prod(shape[1:]))
        bound = (3.0 / fan_in) ** 0.5
        return g.uniform(-bound, bound, shape).astype(np.float32)
    raise ValueError("unknown leaf %r in %r" % (leaf, name))
