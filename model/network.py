"""`model.network` of the reference (model/network.py): same names, served by cofii2p_amd.network.

The class exported here is the strict drop-in: unless the caller asks otherwise (`opt.arithmetic`, or the constructor argument) its
dense contractions are fp32-GRADE - "bf16x6": three bf16 planes per operand, six products, fp32 accumulation; the same error against
fp64 as an exact-fp32 contraction (tools/x6_probe.py, tests/test_forward_gpu.py::test_kitti_frame_bf16x6_is_fp32_grade) - so
`evaluation/eval_all.py` / `train.py` callers that only swap the import get fp32-level results, not the faster but coarser 3-term
split `cofii2p_amd.network.CoFiI2P` defaults to.  `opt.arithmetic = "f32"` selects the exact fp32 MFMA (products bit-equal
to an fmaf chain) at about half the frame rate.  INTEGRATION.md "Arithmetic".  It also replays a hipGraph per input signature without being
asked (results are clones the caller owns): 242 instead of 173 frames/s for an eval_all.py-shaped loop; `enable_graphs(False)` = eager launches."""
from cofii2p_amd import network as _net
from cofii2p_amd.network import (CoFiI2P_wrapper as _Wrapper, extract_patch, fine_matching, fine_process, point2node,  # noqa: F401
                                 score_thresholds, square_distance)


class CoFiI2P(_net.CoFiI2P):
    DEFAULT_ARITHMETIC = "bf16x6"
    DEFAULT_GRAPHS = True   # an unchanged caller gets hipGraph replay (inputs staged, results cloned): same call, same ownership, 1.4x the rate


class CoFiI2P_wrapper(_Wrapper):
    """network.py:267-274 around the strict drop-in class."""

    def __init__(self, opt):
        super(_Wrapper, self).__init__()
        self.cofii2p = CoFiI2P(opt)


__all__ = ["CoFiI2P", "CoFiI2P_wrapper", "fine_process", "extract_patch", "point2node", "square_distance"]
