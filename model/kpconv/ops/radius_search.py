"""model/kpconv/ops/radius_search.py of the reference, served by the HIP kernels."""
from cofii2p_amd.neighbors import radius_search  # noqa: F401
