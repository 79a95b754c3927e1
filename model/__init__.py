"""Import-path shim: `from model.network import CoFiI2P` (evaluation/eval_all.py:11, train.py:13, data/kitti.py:19 of the reference)
resolves to the MI355X implementation when this repository's root is on sys.path instead of (or ahead of) the reference's.
Nothing lives here: every name is re-exported from cofii2p_amd."""
