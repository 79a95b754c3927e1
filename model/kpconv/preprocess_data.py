"""`model.kpconv.preprocess_data` of the reference (imported by data/kitti.py:18, data/nuscenes.py:12): the pyramid builder with
the reference's signature, on the HIP KNN kernels."""
from cofii2p_amd.preprocess import precompute_point_cloud_stack_mode  # noqa: F401

# data/kitti.py:18 imports both names; the reference's second variant (preprocess_data.py:145-203) differs only in WHO searches
# (its torch `knn` instead of open3d's KNNSearch) - the tables are the same KNN-128 pyramid
precompute_point_cloud_cuda = precompute_point_cloud_stack_mode
