"""Row f2 on the GPU: the HIP data-side kernels (csrc/dataside.hip, through the C ABI) against oracle/dataside_oracle.py, which
tests/test_dataside_cpu.py pins against the reference's own __getitem__.  Bit-exact: the kernels and the oracle state the same
float32 / double operation order.  Needs a real MI355X."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cofi_oracle as O  # noqa: E402
import dataside_oracle as D  # noqa: E402
import knn_c  # noqa: E402
from common import load_golden  # noqa: E402
from test_dataside_cpu import INDICES, INT_KEYS, kitti_opt  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ds():
    from cofii2p_amd import dataside

    assert torch.cuda.is_available()
    return dataside


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def calib_P_Tr():
    from cofii2p_amd import synth

    cal = D.calib_matrices(synth.KITTI_CALIB_LINES)
    return np.dot(cal["P2"], cal["Tr"])


def run_voxel(ds, data, P_Tr):
    prep = ds.FramePreparer(kitti_opt(), DEV)
    vox, n = prep.voxel_downsample(G(data), G(P_Tr.astype(np.float32)))
    return vox[:n].cpu().numpy(), n


def oracle_voxel(data, P_Tr):
    pc = D.rigid(P_Tr, data[0:3])
    sn = D.rigid(P_Tr, data[4:], translate=False)
    return D.voxel_down_sample(pc, data[3:4], sn, 0.1), pc, sn


@pytest.mark.parametrize("n_points", [120000, 30000, 1500, 700, 1])
def test_voxel_grid_against_oracle(ds, n_points):
    from cofii2p_amd import synth

    data = synth.make_raw_scan(1, num_points=max(n_points, 16))[0][:, :n_points].copy()
    P_Tr = calib_P_Tr()
    (vp, vi, vs), _, _ = oracle_voxel(data, P_Tr)
    rows, n = run_voxel(ds, data, P_Tr)
    assert n == vp.shape[1]
    assert np.array_equal(rows[:, 0:3], vp.T)
    assert np.array_equal(rows[:, 3], vi[0])
    assert np.array_equal(rows[:, 4:7], vs.T)
    assert not rows[:, 7].any()


def test_voxel_grid_heavy_voxels_and_duplicates(ds):
    """Many points per voxel (long runs in the segmented mean), exact duplicates and points on voxel faces."""
    g = np.random.default_rng(11)
    n = 50000
    pts = g.uniform(-1.0, 1.0, (3, n)).astype(np.float32)          # 20^3 voxels, ~6 points each
    pts[:, 1000:3000] = pts[:, 0:1]                                 # 2000 copies of one point
    pts[:, 3000:4000] = np.round(pts[:, 3000:4000] * 10) / 10       # on the lattice
    data = np.concatenate([pts, g.random((1, n), dtype=np.float32), g.standard_normal((3, n)).astype(np.float32)], 0)
    P = np.identity(4, dtype=np.float32)
    (vp, vi, vs), _, _ = oracle_voxel(data, P)
    rows, cnt = run_voxel(ds, data, P)
    assert cnt == vp.shape[1] < 12000
    assert np.array_equal(rows[:, 0:3], vp.T) and np.array_equal(rows[:, 3], vi[0]) and np.array_equal(rows[:, 4:7], vs.T)


def test_voxel_grid_is_reproducible_and_sorted(ds):
    from cofii2p_amd import synth

    data = synth.make_raw_scan(2)[0]
    P_Tr = calib_P_Tr()
    a, n = run_voxel(ds, data, P_Tr)
    b, m = run_voxel(ds, data, P_Tr)
    assert n == m and np.array_equal(a, b)
    pc = D.rigid(P_Tr, data[0:3]).astype(np.float64)
    minb = pc.min(1) - 0.05
    vid = np.floor((a[:, 0:3].astype(np.float64) - minb) / 0.1).astype(np.int64)
    key = (vid[:, 0] << 26) | (vid[:, 1] << 13) | vid[:, 2]
    assert np.all(np.diff(key) > 0)


def test_voxel_grid_rejects_scans_wider_than_the_key(ds):
    from cofii2p_amd import _lib

    data = np.zeros((7, 64), dtype=np.float32)
    data[0, 1] = 900.0   # 9000 voxels along x
    data[3] = 1.0
    with pytest.raises(_lib.CofiError):
        run_voxel(ds, data, np.identity(4, dtype=np.float32))
    with pytest.raises(_lib.CofiError):
        ds.FramePreparer(kitti_opt(), DEV).voxel_downsample(G(data)[:6], G(np.identity(4, dtype=np.float32)))


def test_gather_transform_against_oracle(ds):
    g = np.random.default_rng(3)
    vox = np.zeros((5000, 8), dtype=np.float32)
    vox[:, :7] = g.standard_normal((5000, 7)).astype(np.float32) * 30
    s = ds.FrameSampler(5)
    choice = s.downsample_choice(5000, 20480)    # fewer rows than num_pc: whole copies + a draw
    P = s.random_transform(kitti_opt())
    prep = ds.FramePreparer(kitti_opt(), DEV)
    pts, feats = prep.resample_transform(G(vox), choice, P)
    sel = vox[choice]
    assert np.array_equal(pts.cpu().numpy(), D.rigid(P, sel[:, 0:3].T).T)
    assert np.array_equal(feats.cpu().numpy()[:, 1:], D.rigid(P, sel[:, 4:7].T, translate=False).T)
    assert np.array_equal(feats.cpu().numpy()[:, 0], sel[:, 3])


@pytest.mark.parametrize("hw", [(376, 1241), (370, 1226), (375, 1242), (160, 512)])
def test_image_resize_crop_against_oracle(ds, hw):
    g = np.random.default_rng(hw[0])
    img = g.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    opt = kitti_opt()
    if hw == (160, 512):   # no resize at all: dst = src, crop = whole image
        rhw, crop = hw, (0, 0)
        small = img
    else:
        _, _, crop, rhw = ds.intrinsics_and_crop(np.eye(3), hw, opt, None)
        small = D.resize_linear_u8(img, rhw[1], rhw[0])
    want = (small[crop[0]:crop[0] + 160, crop[1]:crop[1] + 512].astype(np.float32) / 255.0).transpose(2, 0, 1)
    got = ds.FramePreparer(opt, DEV).image(G(img), rhw, crop).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("index", INDICES)
def test_prepared_frame_against_reference_golden(ds, index):
    """The whole device-side __getitem__ against what the REFERENCE's __getitem__ returned for the same frame index (labels and
    index lists exactly; float tensors to 2e-5, the reference multiplies through BLAS) and against the oracle (exactly)."""
    from cofii2p_amd import synth

    gold = load_golden("dataside_ref.npz")
    tag = "i%d_" % index
    seq_i, cam = gold[tag + "frame_cam"]
    data, img, K = synth.make_raw_scan(int(seq_i))
    if cam == 3:
        img = img[:, ::-1].copy()
    opt = kitti_opt()
    prep = ds.FramePreparer(opt, DEV)
    out = prep.prepare(data, img, K, gold[tag + "P_Tr"], index)
    want = D.prepare_frame(data, img, K, gold[tag + "P_Tr"], index, opt)
    for k in INT_KEYS:
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k]), k
        assert out[k].dtype == (torch.int32 if k in ("fine_xy_coors", "fine_center_kpt_coors") else torch.int64), k
    assert np.array_equal(out["coarse_img_mask"].cpu().numpy(), gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k])
    dd = out["pc_data_dict"]
    assert dd["lengths"] == list(gold[tag + "lengths"])
    np.testing.assert_allclose(dd["points"][4].cpu().numpy(), gold[tag + "points4"], rtol=0, atol=2e-5)
    if index == INDICES[0]:
        np.testing.assert_allclose(dd["points"][0].cpu().numpy(), gold[tag + "points0"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(dd["feats"].cpu().numpy(), gold[tag + "feats"], rtol=0, atol=2e-6)
        assert np.array_equal(np.rint(out["img"].cpu().numpy() * 255).astype(np.uint8), gold[tag + "img"])
    # oracle: exact
    for i in range(5):
        assert np.array_equal(dd["points"][i].cpu().numpy(), want["points"][i])
    assert np.array_equal(dd["feats"].cpu().numpy(), want["feats"])
    assert np.array_equal(out["img"].cpu().numpy(), want["img"])
    assert prep.last["voxels"] == want["voxel"][0].shape[1]
    # the KNN tables of the prepared pyramid: the tie-defined C oracle on the same points (stage 3 and 4: seconds on a CPU)
    for i in (3, 4):
        ref = knn_c.knn_torch_compatible(dd["points"][i].cpu(), dd["points"][i].cpu(), 128)
        assert torch.equal(dd["neighbors"][i].cpu(), ref)
        assert dd["neighbors"][i].dtype == torch.int64


def test_color_jitter_kernel_against_oracle(ds):
    """cofi_color_jitter_chw (train mode, kitti.py:193-201) == oracle/dataside_oracle.py::color_jitter, which tests/test_dataside_cpu.py
    holds to PIL itself: every single operation, drawn orders / factors, clipped blends, the hue wrap, grey pixels - bit for bit."""
    g = np.random.default_rng(2)
    prep = ds.FramePreparer(kitti_opt(), DEV, mode="train")

    def run(img, order, fb, fc, fs, fh):
        x = G(np.ascontiguousarray((img.astype(np.float32) / 255.0).transpose(2, 0, 1)))
        return np.rint(prep.color_jitter(x, order, fb, fc, fs, fh).cpu().numpy().transpose(1, 2, 0) * 255).astype(np.uint8)

    for t in range(24):
        img = g.integers(0, 256, (48, 72, 3), dtype=np.uint8)
        if t % 3 == 0:
            img[:8] = img[:8, :, :1]
        order, fb, fc, fs, fh = D.jitter_params(100 + t)
        assert np.array_equal(run(img, order, fb, fc, fs, fh), D.color_jitter(img, order, fb, fc, fs, fh)), t
        for op in range(4):
            rest = [o for o in range(4) if o != op]
            assert np.array_equal(run(img, [op] + rest, fb, fc, fs, fh), D.color_jitter(img, [op] + rest, fb, fc, fs, fh)), (t, op)
    img = g.integers(0, 256, (160, 512, 3), dtype=np.uint8)
    for f, fh in ((0.0, -0.5), (0.5, -0.004), (1.0, 0.0), (1.7, 0.25), (2.5, 0.5)):
        assert np.array_equal(run(img, [2, 0, 3, 1], f, f, f, fh), D.color_jitter(img, [2, 0, 3, 1], f, f, f, fh)), (f, fh)


def test_prepared_train_frame_against_oracle(ds):
    """FramePreparer(mode='train'): the frame's random crop (kitti.py:312-314, the reference's own `random` stream) and the colour jitter;
    everything else as in val mode.  Against the oracle's prepare_frame(mode='train'), exactly."""
    from cofii2p_amd import synth

    opt = kitti_opt()
    for index in (3, 12):
        data, img, K = synth.make_raw_scan(index)
        out = ds.FramePreparer(opt, DEV, mode="train").prepare(data, img, K, calib_P_Tr(), index)
        want = D.prepare_frame(data, img, K, calib_P_Tr(), index, opt, mode="train")
        val = D.prepare_frame(data, img, K, calib_P_Tr(), index, opt, mode="val")
        assert want["crop"] != val["crop"] and not np.array_equal(want["img"], val["img"])
        assert np.array_equal(out["img"].cpu().numpy(), want["img"])
        for k in ("K", "K_4", "P"):
            assert np.array_equal(out[k].cpu().numpy(), want[k]), k
        for k in INT_KEYS:
            assert np.array_equal(out[k].cpu().numpy(), want[k]), k
        assert np.array_equal(out["pc_data_dict"]["points"][4].cpu().numpy(), want["points"][4])


@pytest.mark.parametrize("index", (0, 3))
def test_prepared_train_frame_against_reference_golden(ds, index):
    """FramePreparer(mode='train') against what the REFERENCE's __getitem__ returned in train mode for the same frame index
    (tests/tools/make_golden_dataside_train.py: crop, intrinsics, labels exactly; the jittered image by its SHA-256)."""
    import hashlib
    from cofii2p_amd import synth

    gold = load_golden("dataside_train_ref.npz")
    tag = "i%d_" % index
    seq_i, cam = gold[tag + "frame_cam"]
    data, img, K = synth.make_raw_scan(int(seq_i))
    if cam == 3:
        img = img[:, ::-1].copy()
    out = ds.FramePreparer(kitti_opt(), DEV, mode="train").prepare(data, img, K, gold[tag + "P_Tr"], index)
    for k in INT_KEYS:
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k]), k
    assert np.array_equal(out["coarse_img_mask"].cpu().numpy(), gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k]), k
    np.testing.assert_allclose(out["pc_data_dict"]["points"][4].cpu().numpy(), gold[tag + "points4"], rtol=0, atol=2e-5)
    q = np.rint(out["img"].cpu().numpy() * 255.0).astype(np.uint8)
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(gold[tag + "img_sha256"])


def test_prepared_frame_feeds_the_model(ds):
    """Loader output -> CoFiI2P forward, as evaluation/eval_all.py:64-83 consumes a sample (batch of one)."""
    from cofii2p_amd import synth
    from cofii2p_amd.network import CoFiI2P
    import bench

    data, img, K = synth.make_raw_scan(0)
    sample = ds.FramePreparer(kitti_opt(), DEV).prepare(data, img, K, calib_P_Tr(), 0)
    model = CoFiI2P(bench.Opt()).to(DEV)
    dd = sample["pc_data_dict"]
    with torch.no_grad():   # train.py:42-70 (validation pass): the loader's labels drive the fine branch
        outs = model(dd, sample["img"][None], sample["fine_center_kpt_coors"], sample["fine_xy_coors"], sample["fine_pc_inline_index"], "val")
        test = model(dd, sample["img"][None], None, None, None, "test")
    assert outs[4].shape[0] == 64 and outs[5].shape[-1] == 64            # num_kpt patches / fine point features
    for o in list(outs) + list(test):
        if torch.is_tensor(o) and o.is_floating_point():
            assert torch.isfinite(o).all()


@pytest.mark.parametrize("index", (0, 1, 2))
def test_prepared_nuscenes_frame_against_reference_golden(ds, index):
    """FramePreparer(dataset='nuscenes') against the reference's nuscenes_pc_img_dataset.__getitem__ (labels and index lists exactly,
    floats to 2e-5) and the oracle (exactly); incl. the sample with valid_kpt False and the cloud with fewer than num_pc points."""
    from cofii2p_amd import synth
    from test_dataside_cpu import nuscenes_opt

    gold = load_golden("dataside_nuscenes_ref.npz")
    tag = "i%d_" % index
    fid, n = gold[tag + "frame_points"]
    pc4, img, K = synth.make_raw_nuscenes(int(fid), int(n))
    opt = nuscenes_opt()
    out = ds.FramePreparer(opt, DEV, dataset="nuscenes").prepare(pc4, img, K, None, index)
    want = D.prepare_frame_nuscenes(pc4, img, K, index, opt)
    assert out["valid_kpt"] == bool(gold[tag + "valid_kpt"])
    for k in INT_KEYS:
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k]), k
    assert np.array_equal(out["coarse_img_mask"].cpu().numpy(), gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k])
    dd = out["pc_data_dict"]
    np.testing.assert_allclose(dd["points"][4].cpu().numpy(), gold[tag + "points4"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dd["feats"].cpu().numpy()[::64], gold[tag + "feats_rows"], rtol=0, atol=2e-5)
    for i in range(5):
        assert np.array_equal(dd["points"][i].cpu().numpy(), want["points"][i])
    assert np.array_equal(dd["feats"].cpu().numpy(), want["feats"])
    assert np.array_equal(out["img"].cpu().numpy(), want["img"]) and tuple(out["img"].shape) == (3, 160, 320)


def test_prepared_nuscenes_train_frame_against_oracle(ds):
    """FramePreparer(dataset='nuscenes', mode='train') (nuscenes.py:232-234 random crop, :249-250 colour jitter) against the oracle, exactly."""
    from cofii2p_amd import synth
    from test_dataside_cpu import nuscenes_opt

    gold = load_golden("dataside_nuscenes_ref.npz")
    fid, n = gold["i1_frame_points"]
    pc4, img, K = synth.make_raw_nuscenes(int(fid), int(n))
    opt = nuscenes_opt()
    for index in (1, 7):
        out = ds.FramePreparer(opt, DEV, dataset="nuscenes", mode="train").prepare(pc4, img, K, None, index)
        want = D.prepare_frame_nuscenes(pc4, img, K, index, opt, mode="train")
        val = D.prepare_frame_nuscenes(pc4, img, K, index, opt)
        assert want["crop"] != val["crop"] and not np.array_equal(want["img"], val["img"])
        assert out["valid_kpt"] and want["valid_kpt"]
        assert np.array_equal(out["img"].cpu().numpy(), want["img"])
        for k in ("K", "K_4", "P"):
            assert np.array_equal(out[k].cpu().numpy(), want[k]), k
        for k in INT_KEYS:
            assert np.array_equal(out[k].cpu().numpy(), want[k]), k
        assert np.array_equal(out["pc_data_dict"]["points"][4].cpu().numpy(), want["points"][4])


@pytest.mark.parametrize("index", (0, 1, 2))
def test_prepared_nuscenes_train_frame_against_reference_golden(ds, index):
    """FramePreparer(dataset='nuscenes', mode='train') against the reference's own train-mode __getitem__
    (tests/tools/make_golden_dataside_nuscenes.py --train), incl. the valid_kpt False sample."""
    import hashlib
    from cofii2p_amd import synth
    from test_dataside_cpu import nuscenes_opt

    gold = load_golden("dataside_nuscenes_train_ref.npz")
    tag = "i%d_" % index
    fid, n = gold[tag + "frame_points"]
    pc4, img, K = synth.make_raw_nuscenes(int(fid), int(n))
    out = ds.FramePreparer(nuscenes_opt(), DEV, dataset="nuscenes", mode="train").prepare(pc4, img, K, None, index)
    assert out["valid_kpt"] == bool(gold[tag + "valid_kpt"])
    for k in INT_KEYS:
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k]), k
    assert np.array_equal(out["coarse_img_mask"].cpu().numpy(), gold[tag + "coarse_img_mask"])
    for k in ("K", "K_4", "P"):
        assert np.array_equal(out[k].cpu().numpy(), gold[tag + k]), k
    np.testing.assert_allclose(out["pc_data_dict"]["points"][4].cpu().numpy(), gold[tag + "points4"], rtol=0, atol=2e-5)
    q = np.rint(out["img"].cpu().numpy() * 255.0).astype(np.uint8)
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(gold[tag + "img_sha256"])


def test_begin_complete_pipeline_equals_prepare(ds):
    """FramePreparer.begin / complete (the voxel grid enqueued ahead, no wait on its count) == prepare, also with two frames interleaved
    on two preparers and the labels deferred."""
    from cofii2p_amd import synth

    opt = kitti_opt()
    P_Tr = calib_P_Tr()
    frames = [synth.make_raw_scan(i) for i in (0, 1)]
    want = [ds.FramePreparer(opt, DEV).prepare(d, im, K, P_Tr, 10 + i) for i, (d, im, K) in enumerate(frames)]
    preps = [ds.FramePreparer(opt, DEV) for _ in frames]
    hs = [p.begin(d, im, K, P_Tr, 10 + i) for i, (p, (d, im, K)) in enumerate(zip(preps, frames))]
    got = [p.complete(h, defer_labels=True) for p, h in zip(preps, hs)]
    for g in got:
        assert "finish_labels" in g and "pc_kpt_idx" not in g
        g["finish_labels"]()
    for w, g in zip(want, got):
        for i in range(5):
            assert torch.equal(w["pc_data_dict"]["points"][i], g["pc_data_dict"]["points"][i])
            assert torch.equal(w["pc_data_dict"]["neighbors"][i], g["pc_data_dict"]["neighbors"][i])
        assert torch.equal(w["img"], g["img"]) and torch.equal(w["pc_data_dict"]["feats"], g["pc_data_dict"]["feats"])
        for k in INT_KEYS + ("coarse_img_mask", "K", "K_4", "P"):
            assert torch.equal(w[k], g[k]), k
