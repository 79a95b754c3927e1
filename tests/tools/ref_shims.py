"""Import WHU-USI3DV/CoFiI2P from /root/reference inside THIS container only.

Test-time harness, never shipped to the GPU box as code that is executed there
(`/root/reference` does not exist on the GPU box).  It installs the three shims
listed in SURVEY.md §8(c):

1. a stub ``open3d`` package (the reference imports it at module import time in
   model/network.py:12, model/kpconv/kernel_points.py:23 and
   model/kpconv/preprocess_data.py:2-3 but the forward path never calls it),
2. ``torch.Tensor.cuda`` -> identity (hard-coded ``.cuda()`` calls at
   model/network.py:105,156,180,181),
3. ``model.kpconv.kpconv.load_kernels`` -> deterministic generator from this
   repository (the original writes a PLY cache into the read-only reference
   tree and needs open3d IO, model/kpconv/kernel_points.py:389-455).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("COFI_REFERENCE_ROOT", "/root/reference")


def have_reference() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "model", "network.py"))


def _stub_open3d():
    if "open3d" in sys.modules:
        return
    o3d = types.ModuleType("open3d")
    for sub in ("geometry", "utility", "io", "ml", "ml.torch", "ml.torch.layers"):
        m = types.ModuleType("open3d." + sub)
        sys.modules["open3d." + sub] = m
        parent = o3d
        parts = sub.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], m)

    class _Unavailable:
        def __init__(self, *a, **k):
            raise RuntimeError("open3d is stubbed in the oracle harness")

    sys.modules["open3d.ml.torch.layers"].KNNSearch = _Unavailable
    sys.modules["open3d.ml.torch.layers"].FixedRadiusSearch = _Unavailable
    sys.modules["open3d"] = o3d


def import_reference():
    """Returns the reference's ``model.network`` module (shimmed)."""
    import torch

    if not have_reference():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    _stub_open3d()
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # shim 3 must be in place before model.kpconv.kpconv binds the name
    import importlib

    kp_mod = importlib.import_module("model.kpconv.kernel_points")
    from cofii2p_amd.weights import kernel_point_table

    def _load_kernels(radius, num_kpoints, dimension, fixed, lloyd=False):
        assert dimension == 3 and fixed == "center"
        return kernel_point_table(num_kpoints, radius)

    kp_mod.load_kernels = _load_kernels
    kpconv_mod = importlib.import_module("model.kpconv.kpconv")
    kpconv_mod.load_kernels = _load_kernels
    return importlib.import_module("model.network")


def reference_options(dataset="kitti"):
    import importlib

    opts = importlib.import_module("data.options")
    return opts.Options_KITTI() if dataset == "kitti" else opts.Options_Nuscenes()
