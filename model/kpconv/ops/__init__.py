"""Import-path shim: `model.kpconv.ops.radius_search` / `grid_subsample` of the reference (which forward to an un-vendored
extension) served by cofii2p_amd.neighbors."""
from cofii2p_amd.neighbors import grid_subsample, radius_search  # noqa: F401
