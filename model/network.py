"""`model.network` of the reference (model/network.py): same names, served by cofii2p_amd.network."""
from cofii2p_amd.network import (CoFiI2P, CoFiI2P_wrapper, extract_patch, fine_matching, fine_process, point2node,  # noqa: F401
                                 score_thresholds, square_distance)

__all__ = ["CoFiI2P", "CoFiI2P_wrapper", "fine_process", "extract_patch", "point2node", "square_distance"]
