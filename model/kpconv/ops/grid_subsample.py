"""model/kpconv/ops/grid_subsample.py of the reference, served by the HIP kernels."""
from cofii2p_amd.neighbors import grid_subsample  # noqa: F401
