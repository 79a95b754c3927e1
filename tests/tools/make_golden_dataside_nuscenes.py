"""Generate tests/golden/dataside_nuscenes_ref.npz by RUNNING THE REFERENCE'S nuscenes_pc_img_dataset.__getitem__ (development container
only):   python tests/tools/make_golden_dataside_nuscenes.py

A synthetic nuScenes-style tree (cofii2p_amd.synth.make_raw_nuscenes: test/{img,ext,int,pc}/<name>.npy) is written to a temporary
directory and read back by the reference's own loader (/root/reference/data/nuscenes.py:177-320).  Stand-ins for calls the image lacks:
cv2.resize -> oracle resize_linear_u8 (parity with cv2 unpinned), open3d.ml KNNSearch -> zeros (KNN is pinned by knn_ref.npz); torchvision
is an empty stub (unused in 'val' mode).  Everything recorded is produced by the reference's own numpy / torch code.

    python tests/tools/make_golden_dataside_nuscenes.py --train    ->  tests/golden/dataside_nuscenes_train_ref.npz

runs the same loader in TRAIN mode (nuscenes.py:232-234 random crop, :249-250 colour jitter); torchvision's ColorJitter is then served by the
oracle's colour jitter with the frame's drawn parameters (pinned to PIL by tests/test_dataside_cpu.py::test_color_jitter_oracle_equals_pil)."""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import ref_shims  # noqa: E402
import dataside_oracle as D  # noqa: E402
from cofii2p_amd import synth  # noqa: E402

# (frame id, points in the stored cloud): more than num_pc, fewer than num_pc (whole copies + a draw), and a sparse one that leaves
# fewer than num_kpt coarse points in the picture for some index (valid_kpt False)
SAMPLES = ((0, 26000), (1, 15000), (2, 21000))


def main():
    train = "--train" in sys.argv
    state = {}
    ref_shims.import_reference()
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    if train:
        from PIL import Image

        class ColorJitterStandIn:   # torchvision.transforms.ColorJitter(brightness, contrast, saturation, hue), nuscenes.py:109-117
            def __init__(self, *ranges):
                assert tuple(tuple(r) for r in ranges) == ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))

            def __call__(self, pil_img):
                state["jitter_calls"] = state.get("jitter_calls", 0) + 1
                return Image.fromarray(D.color_jitter(np.array(pil_img), *D.jitter_params(state["seed"])))

        tv.transforms.ColorJitter = ColorJitterStandIn
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.resize = lambda img, dsize, interpolation=1: D.resize_linear_u8(img, dsize[0], dsize[1])
    sys.modules["cv2"] = cv2
    import importlib

    prep = importlib.import_module("model.kpconv.preprocess_data")

    class KNNStandIn:
        def __init__(self, return_distances=True):
            pass

        def __call__(self, support, query, k):
            return types.SimpleNamespace(neighbors_index=torch.zeros(query.shape[0] * k, dtype=torch.int32))

    prep.KNNSearch = KNNStandIn
    nus = importlib.import_module("data.nuscenes")
    opt = ref_shims.reference_options("nuscenes")
    out = {}
    with tempfile.TemporaryDirectory() as root:
        folder = "train" if train else "test"
        for sub in ("img", "ext", "int", "pc"):
            os.makedirs(os.path.join(root, folder, sub))
        for fid, n in SAMPLES:
            pc4, img, K = synth.make_raw_nuscenes(fid, n)
            name = "%06d.npy" % fid
            np.save(os.path.join(root, folder, "pc", name), pc4)
            np.save(os.path.join(root, folder, "img", name), img)
            np.save(os.path.join(root, folder, "int", name), K)
            np.save(os.path.join(root, folder, "ext", name), np.eye(4))
        opt.data_path = root
        ds = nus.nuscenes_pc_img_dataset(opt, "train" if train else "val")
        assert len(ds) == len(SAMPLES)
        for index, (fid, n) in enumerate(SAMPLES):
            state["seed"], state["jitter_calls"] = index, 0     # nuscenes.py:179-181: seed = index
            r = ds[index]
            assert state["jitter_calls"] == (1 if train else 0)
            tag = "i%d_" % index
            out[tag + "frame_points"] = np.array([fid, n])
            out[tag + "valid_kpt"] = np.array(bool(r["valid_kpt"]))
            for k in ("img", "K", "K_4", "P", "coarse_img_mask", "pc_kpt_idx", "pc_outline_idx", "fine_xy_coors", "coarse_img_kpt_idx",
                      "fine_img_kpt_index", "fine_center_kpt_coors", "coarse_img_outline_index", "fine_pc_inline_index"):
                v = r[k].numpy()
                if k == "img":
                    q = np.rint(v * 255.0)
                    assert np.array_equal((q / 255.0).astype(np.float32), v)
                    out[tag + "img_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(q.astype(np.uint8)).tobytes()).hexdigest())
                    continue
                out[tag + k] = v
            dd = r["pc_data_dict"]
            out[tag + "points4"] = dd["points"][4].numpy()
            out[tag + "feats_rows"] = dd["feats"].numpy()[::64]          # every 64th row of (num_pc, 4)
            out[tag + "points0_rows"] = dd["points"][0].numpy()[::64]
            print(index, "valid_kpt", bool(r["valid_kpt"]), "kpts", len(out[tag + "pc_kpt_idx"]), "mask", int(out[tag + "coarse_img_mask"].sum()))
    path = os.path.join(GOLD, "dataside_nuscenes_train_ref.npz" if train else "dataside_nuscenes_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
