"""cofii2p_amd — MI355X-native (gfx950) implementation of CoFiI2P's coarse-to-fine
image-to-point-cloud correspondence forward path.

Only the hot path of SURVEY.md §8 lives here: the HIP kernels + C-ABI
(`csrc/`, `include/cofi_hip.h`), the ctypes host binding (`_lib.py`, `ops.py`) and the
host-side mirror of the reference interface (`network.py`, `preprocess.py`).
"""
__version__ = "0.1.0"
