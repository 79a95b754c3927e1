"""Generate tests/golden/dataside_train_ref.npz by RUNNING THE REFERENCE'S kitti_pc_img_dataset.__getitem__ IN TRAIN MODE (development
container only).

    python tests/tools/make_golden_dataside_train.py

Same harness and stand-ins as make_golden_dataside.py (open3d voxel grid, cv2.resize, KNNSearch: absent from the image), plus one
more: torchvision is absent, so `transforms.ColorJitter` (data/kitti.py:193-201) is served by the oracle's colour jitter with the
frame's drawn parameters - that operation is pinned separately, bit for bit, to the PIL calls torchvision makes
(tests/test_dataside_cpu.py::test_color_jitter_oracle_equals_pil).  What THIS fixture pins is the reference's own train-mode code
around it: the random crop drawn from `random` (kitti.py:312-314, after the SE(3) draws), the intrinsics of the cropped image, the
point at which the jitter is applied, and every label derived afterwards.  The fixture is data: seeds and expected outputs."""
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

FRAMES = 2
INDICES = (0, 3)   # (frame 0, P2), (frame 1, P3) of sequence 00


def write_tree(root):
    for seq in range(11):
        os.makedirs(os.path.join(root, "calib", "%02d" % seq))
        with open(os.path.join(root, "calib", "%02d" % seq, "calib.txt"), "w") as f:
            for key, text in synth.KITTI_CALIB_LINES.items():
                f.write("%s: %s\n" % (key, text))
        for sub in ("img_P2", "img_P3", "pc_npy_with_normal", "K_P2", "K_P3"):
            os.makedirs(os.path.join(root, "sequences", "%02d" % seq, sub))
    base = os.path.join(root, "sequences", "00")
    for i in range(FRAMES):
        data, img, K = synth.make_raw_scan(i)
        np.save(os.path.join(base, "pc_npy_with_normal", "%06d.npy" % i), data)
        for cam in ("P2", "P3"):
            np.save(os.path.join(base, "img_" + cam, "%06d.npy" % i), img if cam == "P2" else img[:, ::-1].copy())
            np.save(os.path.join(base, "K_" + cam, "%06d.npy" % i), K)


def main():
    from PIL import Image

    ref_shims.import_reference()
    state = {}

    class ColorJitterStandIn:   # torchvision.transforms.ColorJitter(brightness, contrast, saturation, hue)
        def __init__(self, brightness, contrast, saturation, hue):
            assert (tuple(brightness), tuple(contrast), tuple(saturation), tuple(hue)) == ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))

        def __call__(self, pil_img):
            state["jitter_calls"] = state.get("jitter_calls", 0) + 1
            return Image.fromarray(D.color_jitter(np.array(pil_img), *D.jitter_params(state["seed"])))

    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
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
    kitti = importlib.import_module("data.kitti")
    opt = ref_shims.reference_options()
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_tree(root)
        opt.data_path = root
        ds = kitti.kitti_pc_img_dataset(opt, "train")
        assert len(ds) == 2 * FRAMES

        def voxel_standin(pointcloud, intensity, sn, voxel_grid_downsample_size):
            data, P_Tr = state["data"], state["P_Tr"]
            pc_o = D.rigid(P_Tr, data[0:3])
            sn_o = D.rigid(P_Tr, data[4:], translate=False)
            assert np.allclose(pointcloud, pc_o, rtol=0, atol=2e-5) and np.allclose(sn, sn_o, rtol=0, atol=2e-6)
            return D.voxel_down_sample(pc_o, intensity, sn_o, voxel_grid_downsample_size)

        ds.downsample_with_intensity_sn = voxel_standin
        for index in INDICES:
            _, _, _, seq, seq_i, key, _ = ds.dataset[index]
            state["data"] = synth.make_raw_scan(seq_i)[0]
            state["P_Tr"] = np.dot(ds.calibhelper.get_matrix(seq, key), ds.calibhelper.get_matrix(seq, "Tr"))
            (seed,) = np.random.SeedSequence([index]).generate_state(1)   # kitti.py:261-262
            state["seed"], state["jitter_calls"] = int(seed), 0
            r = ds[index]
            assert state["jitter_calls"] == 1
            tag = "i%d_" % index
            out[tag + "frame_cam"] = np.array([seq_i, 2 if key == "P2" else 3])
            out[tag + "P_Tr"] = state["P_Tr"]
            for k in ("img", "K", "K_4", "P", "coarse_img_mask", "pc_kpt_idx", "pc_outline_idx", "fine_xy_coors", "coarse_img_kpt_idx",
                      "fine_img_kpt_index", "fine_center_kpt_coors", "coarse_img_outline_index", "fine_pc_inline_index"):
                v = r[k].numpy()
                if k == "img":
                    q = np.rint(v * 255.0)
                    assert np.array_equal((q / 255.0).astype(np.float32), v)
                    out[tag + "img_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(q.astype(np.uint8)).tobytes()).hexdigest())
                    continue
                out[tag + k] = v
            out[tag + "points4"] = r["pc_data_dict"]["points"][4].numpy()
            print(index, key, "in-picture kpts", len(out[tag + "pc_kpt_idx"]), "K", out[tag + "K"][:2, 2])
    path = os.path.join(GOLD, "dataside_train_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
