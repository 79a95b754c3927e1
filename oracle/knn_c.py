"""ctypes binding of oracle/knn_oracle.c (CPU ORACLE — test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcofi_oracle.so")
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcofi_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.cofi_oracle_knn.restype = ctypes.c_int
        _LIB.cofi_oracle_nearest.restype = ctypes.c_int
    return _LIB


def knn(support: np.ndarray, query: np.ndarray, k: int, return_dist: bool = False):
    support = np.ascontiguousarray(support, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    idx = np.empty((query.shape[0], k), dtype=np.int64)
    dist = np.empty((query.shape[0], k), dtype=np.float32)
    rc = _lib().cofi_oracle_knn(support.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(support.shape[0]),
                                query.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(query.shape[0]), ctypes.c_int(k),
                                idx.ctypes.data_as(ctypes.c_void_p), dist.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return (idx, dist) if return_dist else idx


def nearest(support: np.ndarray, query: np.ndarray) -> np.ndarray:
    support = np.ascontiguousarray(support, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    idx = np.empty((query.shape[0],), dtype=np.int64)
    rc = _lib().cofi_oracle_nearest(support.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(support.shape[0]),
                                    query.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(query.shape[0]),
                                    idx.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return idx


def knn_torch_compatible(support, query, k):
    """Same signature as cofi_oracle.knn_torch (torch tensors in / int64 tensor out)."""
    import torch

    return torch.from_numpy(knn(support.numpy(), query.numpy(), k))
