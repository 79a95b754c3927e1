"""Row f2 (data side of a frame): pins oracle/dataside_oracle.py against the outputs of the REFERENCE's own
kitti_pc_img_dataset.__getitem__ (tests/golden/dataside_ref.npz, tests/tools/make_golden_dataside.py), and checks the host logic
of cofii2p_amd/dataside.py that needs no GPU.  CPU only."""
import hashlib
import types

import numpy as np
import pytest

import dataside_oracle as D
from cofii2p_amd import dataside, synth
from common import load_golden

INDICES = (0, 1, 2)
INT_KEYS = ("pc_kpt_idx", "pc_outline_idx", "fine_xy_coors", "coarse_img_kpt_idx", "fine_img_kpt_index", "fine_center_kpt_coors",
            "coarse_img_outline_index", "fine_pc_inline_index")


def kitti_opt():
    """data/options.py:17-38"""
    return types.SimpleNamespace(img_H=160, img_W=512, num_pc=20480, num_kpt=64, P_tx_amplitude=10, P_ty_amplitude=0, P_tz_amplitude=10,
                                 P_Rx_amplitude=0.0, P_Ry_amplitude=2.0 * np.pi, P_Rz_amplitude=0.0)


@pytest.fixture(scope="module")
def gold():
    return load_golden("dataside_ref.npz")


@pytest.fixture(scope="module")
def oracle_frames(gold):
    out = {}
    for index in INDICES:
        seq_i, cam = gold["i%d_frame_cam" % index]
        data, img, K = synth.make_raw_scan(int(seq_i))
        if cam == 3:
            img = img[:, ::-1].copy()
        out[index] = D.prepare_frame(data, img, K, gold["i%d_P_Tr" % index], index, kitti_opt())
    return out


def test_calibration_matrices_match_the_reference_helper(gold):
    cal = D.calib_matrices(synth.KITTI_CALIB_LINES)
    for key in ("P2", "P3", "Tr"):
        assert np.array_equal(cal[key], gold["calib_" + key])
    assert np.array_equal(np.dot(cal["P2"], cal["Tr"]), gold["i0_P_Tr"])
    assert np.array_equal(dataside.calib_matrices(synth.KITTI_CALIB_LINES)["P3"], gold["calib_P3"])


@pytest.mark.parametrize("index", INDICES)
def test_oracle_labels_match_reference_getitem(gold, oracle_frames, index):
    """Everything downstream of the voxel grid / resize stand-ins: resampling draws, SE(3), sub-sampling draws, intrinsics,
    coarse + fine labels, point2node.  Integer outputs exactly; float outputs to 1e-5 (the reference multiplies through BLAS)."""
    r, tag = oracle_frames[index], "i%d_" % index
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(r[k]), gold[tag + k]), k
        assert np.asarray(r[k]).dtype.kind == "i"
    assert np.array_equal(r["coarse_img_mask"], gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        assert r[k].dtype == np.float32
        np.testing.assert_array_equal(r[k], gold[tag + k])
    np.testing.assert_allclose(r["points"][4], gold[tag + "points4"], rtol=0, atol=2e-5)
    assert [p.shape[0] for p in r["points"]] == list(gold[tag + "lengths"])
    q = np.rint(r["img"] * 255.0).astype(np.uint8)
    if (tag + "img") in gold.files:
        assert np.array_equal(q, gold[tag + "img"])
        assert np.array_equal(r["img"], (gold[tag + "img"].astype(np.float32) / 255.0))
    else:
        assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(gold[tag + "img_sha256"])


@pytest.mark.parametrize("index", (0, 3))
def test_oracle_train_mode_matches_reference_getitem(index):
    """mode='train' against the REFERENCE's own __getitem__ in train mode (tests/tools/make_golden_dataside_train.py): the random crop from
    `random` after the SE(3) draws (kitti.py:312-314), the intrinsics of the cropped image, the jittered image (the ColorJitter stand-in of
    the recording is the oracle's, pinned to PIL below) and every label drawn afterwards."""
    gold = load_golden("dataside_train_ref.npz")
    tag = "i%d_" % index
    seq_i, cam = gold[tag + "frame_cam"]
    data, img, K = synth.make_raw_scan(int(seq_i))
    if cam == 3:
        img = img[:, ::-1].copy()
    r = D.prepare_frame(data, img, K, gold[tag + "P_Tr"], index, kitti_opt(), mode="train")
    v = D.prepare_frame(data, img, K, gold[tag + "P_Tr"], index, kitti_opt())
    assert r["crop"] != v["crop"]
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(r[k]), gold[tag + k]), k
    assert np.array_equal(r["coarse_img_mask"], gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        np.testing.assert_array_equal(r[k], gold[tag + k])
    np.testing.assert_allclose(r["points"][4], gold[tag + "points4"], rtol=0, atol=2e-5)
    q = np.rint(r["img"] * 255.0).astype(np.uint8)
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(gold[tag + "img_sha256"])


def test_oracle_points_and_feats_match_reference(gold, oracle_frames):
    r = oracle_frames[0]
    np.testing.assert_allclose(r["points"][0], gold["i0_points0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(r["feats"], gold["i0_feats"], rtol=0, atol=2e-6)
    assert np.array_equal(r["feats"][:, 0], gold["i0_feats"][:, 0])   # intensity is gathered, not transformed


def test_voxel_grid_properties():
    """Size-independent properties of the voxel grid restatement: every output point lies in its own voxel, voxels are unique and
    ordered, the weighted mean of the outputs is the mean of the inputs, idempotence on one-point-per-voxel input."""
    data = synth.make_raw_scan(3, num_points=30000)[0]
    pc, inten, sn = data[0:3], data[3:4], data[4:]
    vp, vi, vs = D.voxel_down_sample(pc, inten, sn, 0.1)
    minb = pc.astype(np.float64).min(1) - 0.05
    vid = np.floor((vp.T.astype(np.float64) - minb) / 0.1).astype(np.int64)
    key = (vid[:, 0] << 26) | (vid[:, 1] << 13) | vid[:, 2]
    assert np.all(np.diff(key) > 0)
    in_id = np.floor((pc.T.astype(np.float64) - minb) / 0.1).astype(np.int64)
    in_key = (in_id[:, 0] << 26) | (in_id[:, 1] << 13) | in_id[:, 2]
    uk, cnt = np.unique(in_key, return_counts=True)
    assert np.array_equal(uk, key)
    np.testing.assert_allclose((vp.astype(np.float64) * cnt).sum(1) / cnt.sum(), pc.astype(np.float64).mean(1), atol=1e-5)
    assert vi.min() >= inten.min() - 1e-6 and vi.max() <= inten.max() + 1e-6
    # one point per voxel -> the same points back (in key order)
    vp2, vi2, vs2 = D.voxel_down_sample(vp, vi, vs, 0.1)
    if vp2.shape[1] == vp.shape[1]:
        assert np.array_equal(vp2, vp) and np.array_equal(vs2, vs)


def test_resize_exact_half_is_rounded_box_mean():
    """An exact x0.5 bilinear resize reads the 2x2 block with weights 1/4: OpenCV's fixed point gives floor((sum + 2) / 4)."""
    g = np.random.default_rng(5)
    img = g.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    out = D.resize_linear_u8(img, 10, 6)
    blk = img.astype(np.int64).reshape(6, 2, 10, 2, 3).sum((1, 3))
    assert np.array_equal(out, ((blk + 2) >> 2).astype(np.uint8))
    # identity size: untouched
    assert np.array_equal(D.resize_linear_u8(img, 20, 12), img)


def test_downsample_choice_small_cloud_repeats_then_draws():
    """kitti.py:171-176: fewer voxels than num_pc -> whole copies of the cloud plus a draw without replacement."""
    rs = np.random.RandomState(3)
    c = D.downsample_choice(900, 2048, rs)
    assert c.shape == (2048,) and np.array_equal(c[:1800], np.r_[np.arange(900), np.arange(900)])
    assert len(set(c[1800:].tolist())) == 248
    assert np.array_equal(dataside.FrameSampler(0, seed=3).downsample_choice(900, 2048), c)


def test_host_sampler_follows_reference_draw_order(gold, oracle_frames):
    """cofii2p_amd.dataside.FrameSampler (product host code) makes the draws of kitti.py in the reference's order: same choice
    indices, same SE(3), same sub-sampling lists, same label permutations as the oracle that is pinned above."""
    opt = kitti_opt()
    for index in INDICES:
        r = oracle_frames[index]
        s = dataside.FrameSampler(index)
        assert s.seed == D.frame_seed(index)
        assert np.array_equal(s.downsample_choice(r["voxel"][0].shape[1], opt.num_pc), r["choice"])
        assert np.array_equal(s.random_transform(opt), r["P_random"])
        for a, b in zip(s.subsample_indices(opt.num_pc, 5), r["subsample"]):
            assert np.array_equal(a, b)
        K_2, K_4, crop, _ = dataside.intrinsics_and_crop(synth.make_raw_scan(0)[2], (376, 1241), opt, s)
        assert crop == r["crop"]
        lab = dataside.project_labels(r["points"][4], r["P_random"], K_2, K_4, opt, s)
        assert np.array_equal(lab["coarse_img_mask"], gold["i%d_coarse_img_mask" % index])
        for k in INT_KEYS[:-1]:
            assert np.array_equal(lab[k], gold["i%d_%s" % (index, k)]), k


def test_intrinsics_chain(gold):
    K = synth.make_raw_scan(0)[2]
    K_2, K_4, (dy, dx), (rh, rw) = dataside.intrinsics_and_crop(K, (376, 1241), kitti_opt(), None)
    assert (rh, rw) == (188, 620) and (dy, dx) == (14, 54)
    assert np.array_equal(K_2.astype(np.float32), gold["i0_K"]) and np.array_equal(K_4.astype(np.float32), gold["i0_K_4"])


# ---------------------------------------------------------------------------------------------------------------- nuScenes loader
NUS_INDICES = (0, 1, 2)


def nuscenes_opt():
    """data/options.py:60-97"""
    return types.SimpleNamespace(img_H=160, img_W=320, num_pc=20480, num_kpt=32, P_tx_amplitude=10, P_ty_amplitude=0, P_tz_amplitude=10,
                                 P_Rx_amplitude=0.0, P_Ry_amplitude=2.0 * np.pi, P_Rz_amplitude=0.0)


@pytest.mark.parametrize("index", NUS_INDICES)
def test_oracle_matches_reference_nuscenes_getitem(index):
    """data/nuscenes.py:177-320 (seed = index, no voxel grid, features = [intensity | point], valid_kpt) against what the reference's own
    nuscenes_pc_img_dataset.__getitem__ returned (tests/tools/make_golden_dataside_nuscenes.py); index 0 is a sample with fewer than
    num_kpt coarse points in the picture (valid_kpt False, all-zero indices, no permutation drawn)."""
    gold = load_golden("dataside_nuscenes_ref.npz")
    tag = "i%d_" % index
    fid, n = gold[tag + "frame_points"]
    pc4, img, K = synth.make_raw_nuscenes(int(fid), int(n))
    opt = nuscenes_opt()
    r = D.prepare_frame_nuscenes(pc4, img, K, index, opt)
    assert bool(r["valid_kpt"]) == bool(gold[tag + "valid_kpt"])
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(r[k]), gold[tag + k]), k
    assert np.array_equal(r["coarse_img_mask"], gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        np.testing.assert_array_equal(r[k], gold[tag + k])
    np.testing.assert_allclose(r["points"][4], gold[tag + "points4"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(r["points"][0][::64], gold[tag + "points0_rows"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(r["feats"][::64], gold[tag + "feats_rows"], rtol=0, atol=2e-5)
    q = np.rint(r["img"] * 255.0).astype(np.uint8)
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(gold[tag + "img_sha256"])
    # the product's host logic (sampler with the nuScenes seed rule, label projection with the valid_kpt branch)
    s = dataside.FrameSampler(index, dataset="nuscenes")
    assert s.seed == index
    assert np.array_equal(s.downsample_choice(int(n), opt.num_pc), r["choice"])
    assert np.array_equal(s.random_transform(opt), r["P_random"])
    for a, b in zip(s.subsample_indices(opt.num_pc, 5), r["subsample"]):
        assert np.array_equal(a, b)
    K_2, K_4, crop, rhw = dataside.intrinsics_and_crop(K, img.shape[:2], opt, s)
    assert crop == r["crop"] and rhw == (450, 800)
    lab = dataside.project_labels(r["points"][4], r["P_random"], K_2, K_4, opt, s, dataset="nuscenes")
    assert lab["valid_kpt"] == bool(gold[tag + "valid_kpt"])
    for k in INT_KEYS[:-1]:
        assert np.array_equal(lab[k], gold[tag + k]), k


@pytest.mark.parametrize("index", NUS_INDICES)
def test_oracle_train_mode_matches_reference_nuscenes_getitem(index):
    """nuscenes.py:232-234 (random crop: dx, then dy, from `random` after the SE(3) draws) and :249-250 (colour jitter) against the
    reference's own __getitem__ in train mode (tests/tools/make_golden_dataside_nuscenes.py --train), incl. the valid_kpt False sample."""
    gold = load_golden("dataside_nuscenes_train_ref.npz")
    tag = "i%d_" % index
    fid, n = gold[tag + "frame_points"]
    pc4, img, K = synth.make_raw_nuscenes(int(fid), int(n))
    r = D.prepare_frame_nuscenes(pc4, img, K, index, nuscenes_opt(), mode="train")
    assert bool(r["valid_kpt"]) == bool(gold[tag + "valid_kpt"])
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(r[k]), gold[tag + k]), k
    assert np.array_equal(r["coarse_img_mask"], gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        np.testing.assert_array_equal(r[k], gold[tag + k])
    np.testing.assert_allclose(r["points"][4], gold[tag + "points4"], rtol=0, atol=2e-5)
    q = np.rint(r["img"] * 255.0).astype(np.uint8)
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(gold[tag + "img_sha256"])


def test_color_jitter_oracle_equals_pil():
    """train-mode augmentation (kitti.py:193-201): the oracle's restatement of the four ColorJitter operations against PIL itself - what
    torchvision's PIL path calls (ImageEnhance.Brightness / Contrast / Color, the HSV round trip with a uint8 hue shift) - bit for bit,
    incl. grey pixels (hue undefined) and every operation order the frame seeds draw; the product's sampler draws the same parameters."""
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance

    from cofii2p_amd.sampler import FrameSampler

    def pil_jitter(img, order, fb, fc, fs, fh):
        im = Image.fromarray(img)
        for op in order:   # torchvision/transforms/_functional_pil.py: adjust_brightness / _contrast / _saturation / _hue
            if op == 0:
                im = ImageEnhance.Brightness(im).enhance(fb)
            elif op == 1:
                im = ImageEnhance.Contrast(im).enhance(fc)
            elif op == 2:
                im = ImageEnhance.Color(im).enhance(fs)
            else:
                h, s_, v = im.convert("HSV").split()
                np_h = np.array(h, dtype=np.uint8)
                with np.errstate(over="ignore"):
                    np_h += np.array(fh * 255).astype(np.uint8)
                im = Image.merge("HSV", (Image.fromarray(np_h, "L"), s_, v)).convert("RGB")
        return np.array(im)

    g = np.random.default_rng(1)
    orders = set()
    for t in range(48):
        img = g.integers(0, 256, (40, 56, 3), dtype=np.uint8)
        if t % 4 == 0:
            img[:6] = img[:6, :, :1]
        order, fb, fc, fs, fh = D.jitter_params(t)
        orders.add(tuple(order))
        assert (order, fb, fc, fs, fh) == FrameSampler.__new__(FrameSampler).__class__.color_jitter_params(type("S", (), {"seed": t})())
        assert 0.8 <= min(fb, fc, fs) and max(fb, fc, fs) <= 1.2 and -0.1 <= fh <= 0.1
        assert np.array_equal(pil_jitter(img, order, fb, fc, fs, fh), D.color_jitter(img, order, fb, fc, fs, fh)), t
        for op in range(4):
            assert np.array_equal(pil_jitter(img, [op], fb, fc, fs, fh), D.color_jitter(img, [op], fb, fc, fs, fh)), (t, op)
    assert len(orders) > 10
    # extremes of the blend (factor outside [0, 1]: clipped) and of the hue wrap
    img = g.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    for f in (0.0, 0.5, 1.0, 1.7, 2.5):
        for op in (0, 1, 2):
            assert np.array_equal(pil_jitter(img, [op], f, f, f, 0.0), D.color_jitter(img, [op], f, f, f, 0.0)), (op, f)
    for fh in (-0.5, -0.1, -0.004, 0.0, 0.003, 0.25, 0.5):
        assert np.array_equal(pil_jitter(img, [3], 1, 1, 1, fh), D.color_jitter(img, [3], 1, 1, 1, fh)), fh
