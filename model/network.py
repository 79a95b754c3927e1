"""`model.network` of the reference (model/network.py): same names, served by cofii2p_amd.network.

The class exported here is the strict drop-in: unless the caller asks otherwise (`opt.arithmetic = "bf16x3"`, or the constructor
argument) its dense contractions run on the exact fp32 MFMA - the reference's arithmetic - so `evaluation/eval_all.py` / `train.py`
(validation) callers that only swap the import get fp32 products, not the faster 3-term bf16 split `cofii2p_amd.network.CoFiI2P`
defaults to (INTEGRATION.md "Arithmetic")."""
from cofii2p_amd import network as _net
from cofii2p_amd.network import (CoFiI2P_wrapper as _Wrapper, extract_patch, fine_matching, fine_process, point2node,  # noqa: F401
                                 score_thresholds, square_distance)


class CoFiI2P(_net.CoFiI2P):
    DEFAULT_ARITHMETIC = "f32"


class CoFiI2P_wrapper(_Wrapper):
    """network.py:267-274 around the strict drop-in class."""

    def __init__(self, opt):
        super(_Wrapper, self).__init__()
        self.cofii2p = CoFiI2P(opt)


__all__ = ["CoFiI2P", "CoFiI2P_wrapper", "fine_process", "extract_patch", "point2node", "square_distance"]
