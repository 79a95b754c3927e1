"""The pipelined loader (cofii2p_amd/loader.py: voxel grids enqueued ahead, draws in worker processes, resample + pyramid + image as one
hipGraph per slot) against the synchronous FramePreparer.prepare, which tests/test_dataside_gpu.py pins to the reference's own
__getitem__ (tests/golden/dataside_ref.npz): identical bits for every tensor, labels included, with slots reused and frames
interleaved; and PyramidGraph against build_pyramid.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_dataside_cpu import INT_KEYS, kitti_opt  # noqa: E402
from test_dataside_gpu import calib_P_Tr  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("mode", ["val", "train"])
def test_pipelined_loader_equals_prepare(mode):
    """mode='train': the frame's random crop and colour jitter (kitti.py:312-314, 329-330) behind the slot's recorded pyramid"""
    from cofii2p_amd import dataside, synth
    from cofii2p_amd.loader import FrameLoader

    opt, P_Tr = kitti_opt(), calib_P_Tr()
    frames = [synth.make_raw_scan(i) for i in (0, 1)]
    ids = [10, 11, 12, 13, 14, 15, 16]                      # 7 frames through 3 slots: every slot is reused, two raw scans alternate
    want = [dataside.FramePreparer(opt, DEV, mode=mode).prepare(*frames[k % 2], P_Tr, ids[k]) for k in range(len(ids))]
    cap = torch.cuda.Stream(device=DEV)
    loader = FrameLoader(opt, DEV, slots=3, workers=2, capture_stream=cap, mode=mode)
    try:
        got = [None] * len(ids)
        LOOK = 2
        for k in range(min(LOOK, len(ids))):
            loader.begin(k % 3, *frames[k % 2], P_Tr, ids[k])
        for k in range(len(ids)):
            loader.poll()
            smp = loader.complete(k % 3)
            smp["finish_labels"]()
            # the slot's tensors are static: snapshot before the slot is reused
            got[k] = {"img": smp["img"].clone(), "feats": smp["pc_data_dict"]["feats"].clone(),
                      "points": [t.clone() for t in smp["pc_data_dict"]["points"]],
                      "neighbors": [t.clone() for t in smp["pc_data_dict"]["neighbors"]],
                      "subsampling": [t.clone() for t in smp["pc_data_dict"]["subsampling"]],
                      "upsampling": [t.clone() for t in smp["pc_data_dict"]["upsampling"]],
                      **{key: smp[key].clone() if torch.is_tensor(smp[key]) else smp[key] for key in INT_KEYS + ("coarse_img_mask", "K", "K_4", "P")}}
            assert smp["pc_data_dict"]["neighbors"][0].dtype == torch.int32
            loader.release(k % 3)
            if k + LOOK < len(ids):
                loader.begin((k + LOOK) % 3, *frames[(k + LOOK) % 2], P_Tr, ids[k + LOOK])
        torch.cuda.synchronize()
    finally:
        loader.close()
    for w, g in zip(want, got):
        assert torch.equal(w["img"], g["img"]) and torch.equal(w["pc_data_dict"]["feats"], g["feats"])
        for i in range(5):
            assert torch.equal(w["pc_data_dict"]["points"][i], g["points"][i])
            assert torch.equal(w["pc_data_dict"]["neighbors"][i], g["neighbors"][i].long())     # the reference's dtype is int64
        for i in range(4):
            assert torch.equal(w["pc_data_dict"]["subsampling"][i], g["subsampling"][i].long())
            assert torch.equal(w["pc_data_dict"]["upsampling"][i], g["upsampling"][i].long())
        for key in INT_KEYS + ("coarse_img_mask", "K", "K_4", "P"):
            assert torch.equal(w[key], g[key]), key


def test_train_loader_feeds_the_graphed_train_step():
    """data/kitti.py (train mode) -> train.py:186-286 entirely on the device: FrameLoader(mode='train') samples -> batch_from_sample ->
    GraphedTrainStep (first call eager, second recorded, then replays fed from the loader's int32 tables)."""
    from cofii2p_amd import synth
    from cofii2p_amd.loader import FrameLoader
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.train_step import GraphedTrainStep, batch_from_sample
    import bench

    opt, P_Tr = kitti_opt(), calib_P_Tr()
    frames = [synth.make_raw_scan(i) for i in (0, 1)]
    model = CoFiI2P(bench.Opt()).to(DEV)
    optim = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, capturable=True, fused=True)
    step = GraphedTrainStep(model, optim, bench.StepOpt)
    loader = FrameLoader(opt, DEV, slots=2, workers=1, capture_stream=torch.cuda.Stream(device=DEV), mode="train")
    losses = []
    try:
        loader.begin(0, *frames[0], P_Tr, 20)
        for k in range(5):
            loader.poll()
            smp = loader.complete(k % 2)
            if k + 1 < 5:
                loader.begin((k + 1) % 2, *frames[(k + 1) % 2], P_Tr, 21 + k)
            smp["finish_labels"]()
            losses.append(step(*batch_from_sample(smp)).cpu())
            loader.release(k % 2)
    finally:
        loader.close()
    assert step.replays == 4
    L = torch.stack(losses)
    assert torch.isfinite(L).all() and float(L[-1].sum()) < float(L[0].sum())
    model.eval()
    with torch.no_grad():   # the served weights follow the replayed steps
        out = model(*batch_from_sample(smp)[:2], smp["fine_center_kpt_coors"], None, smp["fine_pc_inline_index"], "val")
    assert all(torch.isfinite(t).all() for t in out[:6])


def test_loader_feeds_the_model_in_place():
    """a loader sample goes straight into forward_async(inputs_stable=True) - the graph reads the slot's tensors where they lie - and gives
    what the synchronous path gives"""
    import bench
    from cofii2p_amd import dataside, synth
    from cofii2p_amd.loader import FrameLoader
    from cofii2p_amd.network import CoFiI2P

    opt, P_Tr = kitti_opt(), calib_P_Tr()
    raw, img, K = synth.make_raw_scan(0)
    model = CoFiI2P(bench.Opt()).to(DEV)
    ref_smp = dataside.FramePreparer(opt, DEV).prepare(raw, img, K, P_Tr, 5)
    ref = model(ref_smp["pc_data_dict"], ref_smp["img"][None], None, None, None, "test")
    st = model.frame_streams(1)[0]
    loader = FrameLoader(opt, DEV, slots=2, workers=1, capture_stream=st)
    try:
        for rep in range(2):   # second round replays the captured graphs
            with torch.cuda.stream(st):
                loader.begin(0, raw, img, K, P_Tr, 5)
                smp = loader.complete(0)
                out = model.finish(model.forward_async(0, smp["pc_data_dict"], smp["img"][None], inputs_stable=True))
            for a, b in zip(ref[:6], out[:6]):
                assert torch.equal(a, b)
            assert torch.equal(ref[7], out[7])
            loader.release(0)
    finally:
        loader.close()


def test_loader_frames_into_a_stack_with_deferred_labels():
    """the loader feeding stack-mode submissions: every sample is copied into its row block of a preprocess.FrameStack (the slot is released
    at once and refilled), the labels come from `labels_from` on the stack's copy of the points - equal to the sample's own
    `finish_labels()`; the stack-mode forward over the stack agrees with the per-frame forwards"""
    import copy

    import bench
    from cofii2p_amd import synth
    from cofii2p_amd.loader import FrameLoader
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.preprocess import FrameStack

    opt, P_Tr = kitti_opt(), calib_P_Tr()
    raw, img, K = synth.make_raw_scan(0)
    model = CoFiI2P(bench.Opt()).to(DEV)
    model.enable_graphs(True)
    st = model.frame_streams(1)[0]
    loader = FrameLoader(opt, DEV, slots=2, workers=1, capture_stream=st)
    B, stack, ctxs, want_labels, want_out = 3, None, [], [], []
    try:
        with torch.cuda.stream(st):
            for f in range(B):
                slot = f % 2                      # two slots serve three frames: slot 0 is reused while its frame lives on in the stack
                loader.begin(slot, raw, img, K, P_Tr, 10 + f)
                smp = loader.complete(slot)
                if stack is None:
                    stack = FrameStack(smp["pc_data_dict"], smp["pc_data_dict"]["feats"], smp["img"], B)
                stack.put(f, smp["pc_data_dict"], smp["pc_data_dict"]["feats"], smp["img"])
                ctxs.append(copy.deepcopy(smp["label_ctx"]))   # the sampler inside is a random stream: each label computation consumes its own copy
                want_out.append([t.clone() for t in model.finish(model.forward_async(f, smp["pc_data_dict"], smp["img"][None]))[:6]])
                lab = smp["finish_labels"]()
                want_labels.append({k: lab[k].clone() for k in ("fine_pc_inline_index", "pc_kpt_idx", "pc_outline_idx", "coarse_img_kpt_idx", "fine_xy_coors")})
                loader.release(slot)
                with pytest.raises(Exception):   # the slot's buffers are another frame's from now on
                    smp["finish_labels"]()
            got = model.finish(model.forward_async(8, stack.pyr, stack.img, inputs_stable=True))
        n4, n1 = stack.pyr["points"][-1].shape[0] // B, stack.pyr["points"][1].shape[0] // B
        coarse = stack.pyr["points"][-1].cpu().numpy()
        for f in range(B):
            lab = loader.labels_from(coarse[f * n4:(f + 1) * n4], stack.pyr["points"][1][f * n1:(f + 1) * n1], stack.pyr["points"][-1][f * n4:(f + 1) * n4], ctxs[f])
            for k, v in want_labels[f].items():
                assert torch.equal(lab[k], v), (f, k)
            for a, b in zip(want_out[f][:4], got[f][:4]):   # stack mode runs other launch shapes than a single frame: fp32 rounding apart
                assert float((a - b).abs().max()) < 2e-5, f
    finally:
        loader.close()


def test_pyramid_graph_equals_build_pyramid():
    from cofii2p_amd.preprocess import PyramidGraph, build_pyramid
    from cofii2p_amd.synth import make_frame, subsample_indices

    cap = torch.cuda.Stream(device=DEV)
    pg = PyramidGraph(4096, [2048, 1024, 512, 256], DEV, capture_stream=cap)
    for fid in (3, 4, 5):   # first call captures, the next two replay with other data
        fr = make_frame(fid, 4096)
        pts = torch.from_numpy(fr.points).to(DEV)
        sub = [torch.from_numpy(s).to(DEV) for s in subsample_indices(4096, 5, seed=fid)]
        want = build_pyramid(pts, sub)
        got = pg.run(pts, sub)
        torch.cuda.synchronize()
        for key in ("points", "neighbors", "subsampling", "upsampling"):
            for a, b in zip(want[key], got[key]):
                assert torch.equal(a, b), key
        # `order` is only a PROCESSING order (the cell order of a grid: points of one cell land in it in arrival order, which the
        # counting sort's cursors do not fix): each must be a permutation of its stage, results never depend on it
        for a, b in zip(want["order"], got["order"]):
            assert torch.equal(torch.sort(a)[0], torch.sort(b)[0]) and torch.equal(torch.sort(b)[0].long(), torch.arange(b.numel(), device=DEV))
