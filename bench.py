#!/usr/bin/env python
"""I2P frames/s of the CoFiI2P forward hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun for N > 1; frames shard across ranks, no data-path collective:
"scaling": "weak").  A step = ONE submission of KITTI-shaped synthetic frames (160x512 image, 20 480 points, KNN-128
pyramid already resident in HBM) through `CoFiI2P.forward(mode='test')` + the caller-side fine matching
(evaluation/eval_all.py:99-105) in the fp32-grade arithmetic the reference-named class ships ("bf16x6").  Default: stack-mode
batches of 16 frames (BASELINE.json configs[2]), 4 submissions in flight; `--batch 1` = one frame per submission (configs[1],
also measured by every default run and reported as `batch1_pipeline` / `config.batch1_frames_per_s`).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle = a port of the
reference's forward, timed on a bounded sample on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak (same guide)
HBM_PEAK_GBS = 8000.0


class Opt:
    img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"


def make_inputs(model_dev, frame_ids, num_points):
    from cofii2p_amd.preprocess import build_pyramid
    from cofii2p_amd.synth import make_frame, subsample_indices

    frames = []
    for fid in frame_ids:
        fr = make_frame(fid, num_points=num_points, img_hw=(Opt.img_H, Opt.img_W))
        sub = [torch.from_numpy(s).to(model_dev) for s in subsample_indices(num_points, 5, seed=1000 + fid)]
        pyr = build_pyramid(torch.from_numpy(fr.points).to(model_dev), sub)  # int32 tables, device resident
        pyr["feats"] = torch.from_numpy(fr.feats).to(model_dev)
        frames.append((pyr, torch.from_numpy(fr.img)[None].to(model_dev), fr))
    return frames


def train_labels(pyr, dev, num_kpt=64, seed=0):
    """train.py-shaped labels for a synthetic frame (data/kitti.py:305-420 derives the real ones from the ground-truth pose)."""
    g = np.random.default_rng(seed)
    pts = pyr["points"][-1].cpu().numpy()
    K_4 = np.array([[20.0, 0.0, 32.0], [0.0, 20.0, 10.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    uvw = pts @ K_4.T
    u, v = uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2]
    inside = (pts[:, 2] > 0.5) & (u >= 0) & (u < 64) & (v >= 0) & (v < 20)
    inl, outl = np.nonzero(inside)[0], np.nonzero(~inside)[0]
    kpt = g.choice(inl, num_kpt, replace=len(inl) < num_kpt)
    out = g.choice(outl, num_kpt, replace=len(outl) < num_kpt)
    cidx = (np.clip(np.floor(v[kpt]), 0, 19) * 64 + np.clip(np.floor(u[kpt]), 0, 63)).astype(np.int64)
    ctr = np.stack([g.integers(2, 254, num_kpt), g.integers(2, 78, num_kpt)]).astype(np.int64)
    b = dict(K_4=K_4, P=np.eye(4, dtype=np.float32), pc_kpt_idx=kpt.astype(np.int64), pc_outline_idx=out.astype(np.int64), coarse_img_kpt_idx=cidx,
             fine_center_kpt_coors=ctr, fine_xy=ctr + g.integers(-2, 2, (2, num_kpt)),
             fine_pc_inline_index=g.integers(0, pyr["points"][1].shape[0], num_kpt).astype(np.int64))
    return {k: torch.from_numpy(v_).to(dev) for k, v_ in b.items()}


class StepOpt:
    dist_thres, pos_margin, neg_margin = 1.0, 0.2, 1.8   # data/options.py:39,42-43


def train_step_summary(dev, frame, steps=8, warmup=3, arith="bf16x6", eager=True):
    """ms per optimisation step (device events around forward / backward / optimizer), peak memory, loss trajectory."""
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.train_step import step_losses

    pyr, img, _fr = frame
    model = CoFiI2P(Opt(), arithmetic=arith).to(dev)
    batch = train_labels(pyr, dev)
    optim = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3)   # train.py:163-164
    model.train()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t = [0.0, 0.0, 0.0]
    losses = []
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    for it in range(warmup + steps if eager else 0):
        optim.zero_grad()
        ev[0].record()
        _o, _m, ls = step_losses(model, pyr, img, batch, StepOpt)
        loss = ls[0] + ls[1] + ls[2]
        ev[1].record()
        loss.backward()
        ev[2].record()
        optim.step()
        ev[3].record()
        torch.cuda.synchronize()
        losses.append(round(float(loss.detach()), 4))
        if it >= warmup:
            for j in range(3):
                t[j] += ev[j].elapsed_time(ev[j + 1])
    peak = torch.cuda.max_memory_allocated() - base
    if eager:
        del _o, _m, ls, loss
    # row f3 measured like the forward: every C-ABI call of ONE eager optimisation step (forward + backward go through the same ops.*
    # entry points; the backward's contractions are ops.gemm on transposed operands) is recorded, each family's launches are replayed
    # back to back in a hipGraph and timed with HIP events -> kernel_ms_per_step, and the roofline of the dominant family
    breakdown = None
    try:
        kt = KernelTimer()

        def one():
            optim.zero_grad()
            _o2, _m2, ls2 = step_losses(model, pyr, img, batch, StepOpt)
            (ls2[0] + ls2[1] + ls2[2]).backward()

        kt.record_fn(one)
        per = kt.measure(reps=3)
        per.pop("__f16x3__", None)
        gsum = {"seconds_per_frame": 0.0, "flops_per_frame": 0.0, "launches_per_frame": 0, "f16x3_flops_per_frame": 0.0}
        for n in ("gemm", "gemm_colstats", "gemm_layernorm", "conv2d_nhwc"):
            if n in per:
                for k in gsum:
                    gsum[k] += per[n][k]
        peak_tf, _share = gemm_mix_peak(gsum, arith)   # (the large backward contractions take the f16x3 kernel too)
        breakdown = {"kernel_ms_per_step": {n: round(1e3 * v["seconds_per_frame"], 3) for n, v in sorted(per.items(), key=lambda kv: -kv[1]["seconds_per_frame"])},
                     "recorded_entry_point_calls_per_step": int(sum(v["launches_per_frame"] for v in per.values())),
                     "roofline": {"kernel": "cofi_gemm (forward + backward contractions)", "bound": "mfma", "unit": "TFLOP/s", "peak": peak_tf,
                                  "achieved": gsum["flops_per_frame"] / max(gsum["seconds_per_frame"], 1e-12) / 1e12,
                                  "frac": gsum["flops_per_frame"] / max(gsum["seconds_per_frame"], 1e-12) / 1e12 / peak_tf,
                                  "algorithmic_gflop_per_step": gsum["flops_per_frame"] / 1e9, "calls_per_step": gsum["launches_per_frame"],
                                  "ms_per_step": 1e3 * gsum["seconds_per_frame"]},
                     "note": "one eager step recorded through the ops.* entry points; torch's own element-wise kernels (autograd glue, Adam) are not in these rows"}
        del kt, per
        optim.zero_grad(set_to_none=True)
    except Exception as e:  # noqa: BLE001 - additional information
        breakdown = {"error": "%s: %s" % (type(e).__name__, e)}
    del optim     # the recording below must not find last step's autograd graph alive (its AccumulateGrad nodes sit on this stream)
    # the same step as one hipGraph (cofii2p_amd.train_step.GraphedTrainStep): the eager step is bound by the Python thread issuing ~3 800 launches
    from cofii2p_amd.train_step import GraphedTrainStep

    graphed = None
    try:
        gopt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, capturable=True, fused=True)
        gstep = GraphedTrainStep(model, gopt, StepOpt, validate=False)
        for _ in range(2 + warmup):
            gstep(pyr, img, batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            gl = gstep(pyr, img, batch)
        e1.record()
        torch.cuda.synchronize()
        graphed = {"ms_per_step": e0.elapsed_time(e1) / steps, "steps_per_s": 1e3 * steps / e0.elapsed_time(e1), "loss_after": round(float(gl.sum()), 4),
                   "note": "GraphedTrainStep: forward + losses + backward + Adam recorded once, replayed per frame (static inputs restaged every call)"}
        del gstep, gopt
    except Exception as e:  # noqa: BLE001 - an optional leg: the eager figure stands on its own
        graphed = {"error": "%s: %s" % (type(e).__name__, e)}
    del model
    torch.cuda.empty_cache()
    return {"ms_per_step": sum(t) / steps, "forward_ms": t[0] / steps, "backward_ms": t[1] / steps, "optimizer_ms": t[2] / steps, "steps": steps,
            "graphed": graphed, "breakdown": breakdown,
            "arithmetic": arith, "num_kpt": 64, "peak_mem_GB": peak / 2 ** 30, "loss": losses,
            "note": "train.py:186-286 on the bench frame: model.train(); forward(mode='train'); desc / overlap / fine-circle losses; backward; Adam. "
                    "HIP kernels in both directions for every weight contraction, KPConv aggregation, attention and neighbour gather "
                    "(cofii2p_amd/autograd.py); not part of `value`"}


def record_kernel_calls(model, dev, points=20480, batch=1):
    """KernelTimer holding every C-ABI call of one submission (one frame, or a stack-mode batch) - used by tools/."""
    from cofii2p_amd.network import CoFiI2P

    frames = make_inputs(dev, list(range(batch)), points)
    one_step(model, frames[0])
    model.enable_graphs(False)
    kt = KernelTimer()
    if batch == 1:
        kt.record(model, frames[0])
    else:
        pyr_b, img_b = CoFiI2P.stack_frames([f[0] for f in frames], [f[1] for f in frames])
        P_b = model._pack(dev)
        kt.record_fn(lambda: model._run_device(P_b, pyr_b["points"], pyr_b["neighbors"], pyr_b["subsampling"], pyr_b["upsampling"],
                                               pyr_b["feats"], img_b, "test", None, None))
    return kt


def graph_node_census(run):
    """GPU dispatches of one submission, counted where the GPU sees them: `run` is captured into a hipGraph (as the product path does) and
    the graph's nodes are counted by type through hipGraphGetNodes / hipGraphNodeGetType.  Entry-point calls (KernelTimer) under-count:
    one cofi_* call may launch several kernels (split-K fold, statistics finalize), and torch's own copies are invisible to it."""
    import ctypes

    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        run()
    raw = g.raw_cuda_graph()
    cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), "libamdhip64.so"]
    hip = None
    for c in cand:
        try:
            hip = ctypes.CDLL(c)
            break
        except OSError:
            continue
    if hip is None:
        return None
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
        return None
    nodes = (ctypes.c_void_p * n.value)()
    hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n))
    names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "child_graph", 5: "empty", 6: "wait_event", 7: "event_record"}
    census = {}
    for nd in nodes:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
        k = names.get(t.value, "type_%d" % t.value)
        census[k] = census.get(k, 0) + 1
    census["dispatches"] = sum(v for k, v in census.items() if k in ("kernel", "memcpy", "memset"))
    del g
    return census


def submission_census(model, dev, frames_list):
    """graph_node_census of the device part of one forward(mode='test') over `frames_list` (one frame, or a stack-mode batch)."""
    from cofii2p_amd.network import CoFiI2P

    pyr_b, img_b = CoFiI2P.stack_frames([f[0] for f in frames_list], [f[1] for f in frames_list])
    P_b = model._pack(dev)
    with torch.no_grad():
        return graph_node_census(lambda: model._run_device(P_b, pyr_b["points"], pyr_b["neighbors"], pyr_b["subsampling"], pyr_b["upsampling"],
                                                           pyr_b["feats"], img_b, "test", None, None))


_model_ref = []


def make_streams(dev, n):
    """n HIP streams to keep frames in flight on.  The model hands out the streams the process already owns first (capture + default
    stream): every extra live stream makes two frame streams share one of HIP's 4 hardware queues (CoFiI2P.frame_streams)."""
    return _model_ref[0].frame_streams(n, dev) if _model_ref else [torch.cuda.Stream(device=dev) for _ in range(n)]


def one_step(model, frame):
    """forward(mode='test') computes the coarse matches, the 4x4 patches AND (in the same hipGraph) the
    caller-side fine matching of eval_all.py:99-105; the returned tuple is the reference's 8-tuple."""
    pyr, img, _ = frame
    out = model(pyr, img, None, None, None, "test")
    return out, model.last_match["fine_xy"]


class KernelTimer:
    """Per-entry-point kernel time of one frame, measured with HIP events on the launch stream.

    Pass 1 (eager) records every C-ABI call of one forward with its live argument tensors and its
    algorithmic work.  Pass 2 replays, per entry point, exactly those launches back-to-back inside one
    hipGraph and times the replay with events: no CPU launch gaps, the same shapes/data as the frame."""

    NAMES = ("gemm", "gemm_colstats", "gemm_layernorm", "conv2d_nhwc", "loftr_tail", "group_stats_from_colpart", "col_inv_norm_from_colpart", "kpconv_aggregate", "attention_parts", "neighbor_maxpool", "group_stats", "group_norm_apply", "layer_norm", "l2norm_rows",
             "gather_rows", "row_sum_positive", "col_inv_norm")

    def __init__(self):
        self.calls = {}

    @staticmethod
    def work(name, a, k):
        if name in ("gemm", "gemm_colstats", "gemm_layernorm"):
            M, K = a[0].shape
            N = a[1].shape[0]
            return 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)
        if name == "loftr_tail":
            L = a[0].shape[0]
            pn = sum(pw.shape[1] for pw, _y, _p in k.get("proj", ()))   # fused projections of the following layers
            return 2.0 * L * (128 * 128 + 256 * 256 + 256 * 128 + 128 * pn), 4.0 * (3 * L * 128 + 128 * 128 + 256 * 256 + 256 * 128 + pn * (128 + L))
        if name == "conv2d_nhwc":
            x, H, W, w, ks = a[0], a[1], a[2], a[3], a[4]
            stride = a[5] if len(a) > 5 else k.get("stride", 1)
            pad = a[6] if len(a) > 6 else k.get("pad", 1)
            Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
            M, N, K = k.get("frames", 1) * Ho * Wo, w.shape[0], w.shape[1]
            return 2.0 * M * N * K, 4.0 * (x.numel() + w.numel() + M * N)
        if name == "kpconv_aggregate":
            feats, idx = a[0], a[3]
            M, H = idx.shape
            C = feats.shape[1]
            # algorithmic bytes: every source row once + index table + output (the gather itself must come from cache)
            return 2.0 * M * H * 15 * C, 4.0 * (feats.shape[0] * C + M * H + M * 15 * C)
        if name == "attention_parts":
            L, HD = a[0].shape
            S = a[1].shape[0]
            return 4.0 * L * S * HD / k.get("frames", 1), 4.0 * (2 * L * HD + 2 * S * HD)
        if name == "neighbor_maxpool":
            x, idx = a[0], a[1]
            return 0.0, 4.0 * (x.numel() + idx.numel() + idx.shape[0] * x.shape[1])
        return 0.0, 8.0 * a[0].numel()

    _depth = 0

    def record(self, model, frame):
        self.record_fn(lambda: one_step(model, frame))

    def record_fn(self, run):
        from cofii2p_amd import ops

        orig = {}
        for name in self.NAMES:
            fn = getattr(ops, name)
            orig[name] = fn

            def rec(*a, _n=name, _f=fn, **k):
                if self._depth:  # an entry point called from inside another recorded wrapper (e.g. the statistics
                    return _f(*a, **k)  # finalize inside group_norm_apply) is timed with its caller, once
                kk = {x: y for x, y in k.items() if x != "out"}  # replay into fresh outputs
                self.calls.setdefault(_n, []).append((_f, a, kk, self.work(_n, a, k)))
                self._depth += 1
                try:
                    return _f(*a, **k)
                finally:
                    self._depth -= 1

            setattr(ops, name, rec)
        try:
            run()
        finally:
            for name, fn in orig.items():
                setattr(ops, name, fn)
        torch.cuda.synchronize()

    def measure(self, reps=5):
        import ctypes

        from cofii2p_amd import _lib

        census = _lib.load().cofi_tune_f16x3_launch_flops   # include/cofi_hip_tune.h: which of this thread's contractions took the f16x3 kernel
        census.argtypes, census.restype = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)], ctypes.c_long
        def timed(calls):
            def run():
                for fn, a, k, _ in calls:
                    fn(*a, **k)
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            del g
            return s.elapsed_time(e) * 1e-3 / reps

        out = {}
        f16_calls = []   # the contraction launches (all entry points) that ran on gemm_f16_big_kernel: timed once more on their own
        for name, calls in self.calls.items():
            f16_flops, f16_n = 0.0, 0
            if name in ("gemm", "gemm_colstats", "conv2d_nhwc"):
                for c in calls:   # one eager call each: did it take the f16x3 kernel?
                    census(1, None)
                    c[0](*c[1], **c[2])
                    fl = ctypes.c_double(0.0)
                    if census(1, ctypes.byref(fl)) > 0:
                        f16_n += 1
                        f16_flops += fl.value
                        f16_calls.append(c)
            sec = timed(calls)
            out[name] = {"seconds_per_frame": sec, "flops_per_frame": sum(c[3][0] for c in calls), "bytes_per_frame": sum(c[3][1] for c in calls),
                         "launches_per_frame": len(calls), "f16x3_flops_per_frame": f16_flops, "f16x3_launches": f16_n}
        if f16_calls:
            out["__f16x3__"] = {"seconds_per_frame": timed(f16_calls), "flops_per_frame": sum(c[3][0] for c in f16_calls),
                                "bytes_per_frame": sum(c[3][1] for c in f16_calls), "launches_per_frame": len(f16_calls),
                                "f16x3_flops_per_frame": sum(c[3][0] for c in f16_calls), "f16x3_launches": len(f16_calls)}
        return out


def kernel_rooflines(model, dev, args, Bsz, frame=None, batch=None):
    """roofline / roofline_attention / kernel_ms_per_frame of one submission (a frame, or a stack-mode batch of Bsz frames): every
    C-ABI call of the submission is recorded, then each entry point's launches are replayed back-to-back in a hipGraph and timed
    with HIP events on the launch stream (KernelTimer)."""
    out = {}
    kt = KernelTimer()
    if Bsz == 1:
        kt.record(model, frame)
    else:
        pyr_b, img_b = batch
        P_b = model._pack(dev)
        kt.record_fn(lambda: model._run_device(P_b, pyr_b["points"], pyr_b["neighbors"], pyr_b["subsampling"], pyr_b["upsampling"],
                                               pyr_b["feats"], img_b, "test", None, None))
    # cross-attention launches (frames == Bsz: one stream attends to the other) apart from the joint self-attention ones (2 Bsz)
    att = kt.calls.get("attention_parts", [])
    cross = [c for c in att if c[2].get("frames", 1) == Bsz]
    if cross and len(cross) != len(att):
        kt.calls["attention_cross"] = cross
    per = kt.measure()
    del kt
    for v in per.values():  # per FRAME figures (launch counts stay per submission)
        for kk in ("seconds_per_frame", "flops_per_frame", "bytes_per_frame", "f16x3_flops_per_frame"):
            v[kk] /= Bsz
        v["launches_per_frame"] = v["launches_per_frame"] / Bsz
    across = per.pop("attention_cross", None)
    f16only = per.pop("__f16x3__", None)
    if f16only and f16only["seconds_per_frame"] > 0:
        # the launches of the dominant KERNEL alone (gemm_f16_big_kernel + its repair launch): three fp16 products per fp32 product -> 2500 / 3
        ach = f16only["flops_per_frame"] / f16only["seconds_per_frame"] / 1e12
        out["roofline_f16x3_kernel"] = {"kernel": "gemm_f16_big_kernel (the cofi_gemm_f32* / cofi_conv2d_nhwc launches that run on it)", "bound": "mfma",
                                        "achieved": ach, "peak": BF16_MFMA_PEAK_TF / 3.0, "unit": "TFLOP/s", "frac": ach / (BF16_MFMA_PEAK_TF / 3.0),
                                        "launches_per_frame": f16only["launches_per_frame"], "ms_per_frame": 1e3 * f16only["seconds_per_frame"],
                                        "avg_launch_us": 1e6 * f16only["seconds_per_frame"] * Bsz / max(1, f16only["launches_per_frame"] * Bsz),
                                        "algorithmic_gflop_per_frame": f16only["flops_per_frame"] / 1e9}
    # the GEMM / implicit-GEMM convolution entry points launch the same MFMA kernel (different loaders / epilogues): one roofline row
    gsum = {"seconds_per_frame": 0.0, "flops_per_frame": 0.0, "bytes_per_frame": 0.0, "launches_per_frame": 0, "f16x3_flops_per_frame": 0.0, "f16x3_launches": 0}
    for n in ("gemm", "gemm_colstats", "gemm_layernorm", "conv2d_nhwc"):
        if n in per:
            for k in gsum:
                gsum[k] += per[n][k]
            del per[n]
    per["gemm"] = gsum
    dom = max(per, key=lambda n: per[n]["seconds_per_frame"])
    d = per[dom]
    if dom != "gemm" and per["gemm"]["seconds_per_frame"] > 0:
        # the contraction family next to a dominant attention kernel (the stress configuration): its own row, counter traffic from the
        # committed PMC passes of that configuration when they exist
        gd = per["gemm"]
        gpeak, gshare = gemm_mix_peak(gd, args.gemm)
        gach = gd["flops_per_frame"] / gd["seconds_per_frame"] / 1e12
        gp = pmc_traffic("gemm", "pmc_traffic_stress.json") if (args.gemm == "bf16x6" and args.points == 40960 and Opt.img_H == 896 and Bsz == 1) else None
        out["roofline_gemm"] = {"kernel": "cofi_gemm", "bound": "mfma", "achieved": gach, "peak": gpeak, "unit": "TFLOP/s", "frac": gach / gpeak,
                                "f16x3_flop_share": gshare, "f16x3_launches_per_submission": gd.get("f16x3_launches", 0),
                                "traffic": None if gp is None else gp.get("traffic_bytes_per_launch"),
                                "traffic_collected_on": None if gp is None else gp.get("collected_on"),
                                "algorithmic_bytes_per_launch": gd["bytes_per_frame"] / gd["launches_per_frame"],
                                "launches_per_frame": gd["launches_per_frame"], "avg_launch_us": 1e6 * gd["seconds_per_frame"] / gd["launches_per_frame"],
                                "algorithmic_gflop_per_frame": gd["flops_per_frame"] / 1e9, "ms_per_frame": 1e3 * gd["seconds_per_frame"]}
    if d["flops_per_frame"] > 0:
        ach = d["flops_per_frame"] / d["seconds_per_frame"] / 1e12
        # the bf16-split GEMM issues 3 bf16 MFMA flops per algorithmic flop: its MFMA roof for ALGORITHMIC flops is 2500/3
        # ... and the 6-term split 6: 2500 / 6
        peak = attention_peak_tf() if dom == "attention_parts" else FP32_MFMA_PEAK_TF
        mix_share = None
        if dom == "gemm" and args.gemm in ("bf16x3", "bf16x6"):
            peak, mix_share = gemm_mix_peak(d, args.gemm)
        # committed PMC passes exist for the two default pipelines in the fp32-grade arithmetic: stack-mode batches of 16 and batch 1
        pmc = None
        if args.gemm == "bf16x6" and args.points == 20480 and Opt.img_H == 160 and Bsz in (1, 16):
            pmc = pmc_traffic(dom, "pmc_traffic.json" if Bsz == 16 else "pmc_traffic_batch1.json")
        elif args.gemm == "bf16x6" and args.points == 40960 and Opt.img_H == 896 and Bsz == 1 and dom == "gemm":
            pmc = pmc_traffic(dom, "pmc_traffic_stress.json")   # the stress configuration (its contractions outweigh the attention kernel now)
        out["roofline"] = {"kernel": "cofi_" + dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                           # HBM bytes per launch from the committed rocprofv3 PMC passes of the default command (stamped with
                           # the commit they were collected on); None for any other configuration
                           "traffic": None if pmc is None else pmc.get("traffic_bytes_per_launch"),
                           "traffic_collected_on": None if pmc is None else pmc.get("collected_on"),
                           "algorithmic_bytes_per_launch": d["bytes_per_frame"] / d["launches_per_frame"],
                           "launches_per_frame": d["launches_per_frame"], "avg_launch_us": 1e6 * d["seconds_per_frame"] / d["launches_per_frame"],
                           "algorithmic_gflop_per_frame": d["flops_per_frame"] / 1e9}
        if mix_share is not None:
            out["roofline"].update({"f16x3_flop_share": mix_share, "f16x3_launches_per_submission": d.get("f16x3_launches", 0),
                                    "peak_note": "roof of the launch MIX for algorithmic flops: 2500 / 6 TF/s for the six-product bf16 kernels, 2500 / 3 for the "
                                                 "launches on the three-product fp16 kernel (f16x3_flop_share of the flops); total flops / sum of each part's time at its roof"})
    else:
        ach = d["bytes_per_frame"] / d["seconds_per_frame"] / 1e9
        out["roofline"] = {"kernel": "cofi_" + dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None}

    def att_row(a, what):
        ach = a["flops_per_frame"] / a["seconds_per_frame"] / 1e12
        from cofii2p_amd import ops as _ops

        return {"kernel": "cofi_attention_parts", "launches": what, "arithmetic": _ops.attention_arith(), "bound": "mfma", "achieved": ach,
                "peak": attention_peak_tf(), "unit": "TFLOP/s", "frac": ach / attention_peak_tf(), "launches_per_frame": a["launches_per_frame"],
                "avg_launch_us": 1e6 * a["seconds_per_frame"] * Bsz / (a["launches_per_frame"] * Bsz)}

    if per.get("attention_parts"):
        per["attention"] = per.pop("attention_parts")   # the attention KERNEL (cofi_attention_parts); the slot merge is the consumer's
        out["roofline_attention"] = att_row(per["attention"], "all (cross + joint self)")
    if across:
        out["roofline_cross_attention"] = att_row(across, "cross-attention only")
    out["kernel_ms_per_frame"] = {n: round(1e3 * v["seconds_per_frame"], 4) for n, v in sorted(per.items(), key=lambda kv: -kv[1]["seconds_per_frame"])}
    out["launches_per_frame"] = {n: v["launches_per_frame"] for n, v in per.items()}
    return out


def gemm_mix_peak(gd, gemm_mode):
    """Matrix roof of a contraction launch list for ALGORITHMIC flops.  bf16x3: 2500 / 3; bf16x6: 2500 / 6 for the launches of the six-product
    kernels and 2500 / 3 for those that took the three-product fp16 kernel (COFI_GEMM_F16X3: fp16 and bf16 MFMA share the 2500 TF/s dense
    peak) - the roof of the MIX is total flops over the sum of each part's time at its own roof.  -> (peak TF/s, share of the flops on f16x3)"""
    if gemm_mode == "f32":
        return FP32_MFMA_PEAK_TF, 0.0
    if gemm_mode == "bf16x3":
        return BF16_MFMA_PEAK_TF / 3.0, 0.0
    tot, f16 = gd["flops_per_frame"], min(gd.get("f16x3_flops_per_frame", 0.0), gd["flops_per_frame"])
    if tot <= 0:
        return BF16_MFMA_PEAK_TF / 6.0, 0.0
    t = f16 / (BF16_MFMA_PEAK_TF / 3.0) + (tot - f16) / (BF16_MFMA_PEAK_TF / 6.0)
    return tot / t, f16 / tot


def attention_peak_tf():
    """Matrix-pipe roof of the attention kernel for ALGORITHMIC flops in the arithmetic it runs in (ops.attention_arith): the 6-term bf16
    split issues six bf16 MFMA flops per algorithmic flop (2500 / 6), the exact kernel runs on the fp32 instruction (157.3)."""
    from cofii2p_amd import ops

    return BF16_MFMA_PEAK_TF / 6.0 if ops.attention_arith() == "bf16x6" else FP32_MFMA_PEAK_TF


def pmc_traffic(kernel_family, fname="pmc_traffic.json"):
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, produced by tools/pmc_to_json.py from separate FETCH_SIZE / WRITE_SIZE runs of this
    same command; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  -> the family's record with the commit the
    passes were collected on, or None if absent."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    try:
        doc = json.load(open(path))
        rec = doc.get(kernel_family)
        if rec is None:
            return None
        rec = dict(rec)
        rec["collected_on"] = doc.get("collected_on", "round 1 (before the commit stamp existed)")
        return rec
    except Exception:
        return None


def stress_summary(dev, args):
    """BASELINE configs[4] (large-attention / HBM-bound stress): 896 x 1600 image (22 400 image tokens; 900 is not divisible by 32, which
    the reference's up-sampler needs - SURVEY.md section 7), 40 960 points, one frame, reference-surface forward + fine matching."""
    from cofii2p_amd.network import CoFiI2P

    saved = (Opt.img_H, Opt.img_W)
    Opt.img_H, Opt.img_W = 896, 1600
    try:
        big = CoFiI2P(Opt()).to(dev)
        big.enable_graphs(True)
        fr = make_inputs(dev, [0], 40960)[0]
        for _ in range(2):
            one_step(big, fr)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)   # peak_mem_GB below is this configuration's own peak, not the bench process's so far
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            out, _ = one_step(big, fr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res = {"workload": "896x1600 image (22400 image tokens), 40960 points (2560 point tokens), batch 1, forward(mode='test') + fine matching, "
                           "one frame at a time", "ms_per_frame": 1e3 * dt, "frames_per_s": 1.0 / dt, "matches": int(out[4].shape[0]),
               "peak_mem_GB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
        if not args.no_kernel_timing:
            big.enable_graphs(False)
            saved_pts, args.points = args.points, 40960
            res.update(kernel_rooflines(big, dev, args, 1, frame=fr))
            args.points = saved_pts
        del big
        return res
    finally:
        Opt.img_H, Opt.img_W = saved


def cpu_baseline(frame, n_frames=2):
    """The CPU oracle (port of the reference forward, validated against reference-generated golden
    vectors) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cofi_oracle as O

    from cofii2p_amd.spec import synth_state_dict

    pyr, img, fr = frame
    data = {k: ([t.cpu().long() if t.dtype in (torch.int32, torch.int64) else t.cpu() for t in v] if isinstance(v, list) and torch.is_tensor(v[0])
                else (v.cpu() if torch.is_tensor(v) else v)) for k, v in pyr.items()}
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict().items()}
    best = None
    ncpu = os.cpu_count() or 8
    with torch.no_grad():
        for threads in sorted({min(8, ncpu), min(32, ncpu)}):
            torch.set_num_threads(threads)
            O.forward(sd, data, img.cpu(), None, None, "test")  # warm-up
            t0 = time.time()
            for _ in range(n_frames):
                res = O.forward(sd, data, img.cpu(), None, None, "test")
                O.fine_match(res[4], res[5], res[6])
            fps = n_frames / (time.time() - t0)
            if best is None or fps > best[0]:
                best = (fps, threads)
    return {"value": best[0], "unit": "frames/s", "cores": best[1], "kind": "port",
            "sample": "%d KITTI-shaped frames (forward test mode + fine match) through oracle/cofi_oracle.py, torch-CPU fp32, KNN pyramid "
                      "precomputed; best of 8 / 32 torch threads on a %d-core host" % (n_frames, ncpu)}


def pin_to_gpu_numa_node(local: int):
    """Best effort: restrict this rank's host threads to the NUMA node its GPU hangs off (8-GPU nodes: a rank whose launch thread sits on
    the other socket pays for every submission).  -> the node id, or None when sysfs does not say."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, set(cpus))
        return node
    except Exception:
        return None


class Pipeline:
    """The software pipeline of the headline measurement: S frame streams x `slots_per_stream` hipGraph slots; frame i is enqueued on stream
    i % S while the previous frames are still executing, a slot's result (incl. the host read of the match count) is collected just
    before the slot is reused.  Every step is ONE frame through the complete forward + fine matching."""

    def __init__(self, model, dev, frames, S, slots_per_stream, copy_inputs):
        self.model, self.frames, self.S, self.copy_inputs = model, frames, S, copy_inputs
        self.streams = make_streams(dev, S)
        self.NSLOT = S * max(1, slots_per_stream)   # submission i: stream i % S, hipGraph slot i % NSLOT
        self.pending = [None] * self.NSLOT

    def run(self, nsteps, base=0):
        nm, model = 0, self.model
        for i in range(nsteps):
            sl = i % self.NSLOT
            if self.pending[sl] is not None:
                nm = model.finish(self.pending[sl])[4].shape[0]
            pyr, img, _ = self.frames[(base + i) % len(self.frames)]
            with torch.cuda.stream(self.streams[i % self.S]):
                self.pending[sl] = model.forward_async(sl, pyr, img, inputs_stable=not self.copy_inputs)
        for k in range(self.NSLOT):   # collect in submission order
            sl = (nsteps + k) % self.NSLOT
            if self.pending[sl] is not None:
                nm = model.finish(self.pending[sl])[4].shape[0]
                self.pending[sl] = None
        return nm

    def warm(self, warmup):
        import math

        # every (slot, input set) graph captured and replayed at least once, whatever --warmup says (in-place inputs: one graph per pair)
        return self.run(max(warmup, 2 * self.NSLOT, math.lcm(self.NSLOT, len(self.frames))), 0)


def _nccl_version():
    """version of the collective library behind torch's "nccl" backend (RCCL on ROCm), or None when torch cannot say"""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:   # noqa: BLE001 - informational
        return None


class BatchPipeline:
    """Stack-mode submissions: `Bsz` frames per submission through the same launches (CoFiI2P.stack_frames), `S` submissions in flight on S
    frame streams, one hipGraph slot per stream; two distinct batches alternate.  A step = one submission."""

    def __init__(self, model, dev, frames, Bsz, S, copy_inputs, slot_base=0):
        from cofii2p_amd.network import CoFiI2P

        self.model, self.S, self.Bsz, self.copy_inputs, self.slot_base = model, S, Bsz, copy_inputs, slot_base
        self.batches = []
        for b in range(2):
            grp = [frames[(b * Bsz + i) % len(frames)] for i in range(Bsz)]
            self.batches.append(CoFiI2P.stack_frames([g[0] for g in grp], [g[1] for g in grp]))
        self.streams = make_streams(dev, S)
        self.pending = [None] * S

    def run(self, nsteps, base=0):
        nm, model = 0, self.model
        for i in range(nsteps):
            sl = i % self.S
            if self.pending[sl] is not None:
                nm = model.finish(self.pending[sl])[0][4].shape[0]
            pyr, img = self.batches[i % len(self.batches)]
            with torch.cuda.stream(self.streams[sl]):
                self.pending[sl] = model.forward_async(self.slot_base + sl, pyr, img, inputs_stable=not self.copy_inputs)
        for sl in range(self.S):
            if self.pending[sl] is not None:
                nm = model.finish(self.pending[sl])[0][4].shape[0]
                self.pending[sl] = None
        return nm

    def warm(self, warmup):
        return self.run(max(warmup, 2 * self.S * len(self.batches)))


def dataside_inputs(dev, points):
    """raw KITTI-shaped scan + image resident in HBM, the calibration, and the options of the device-side loader (row f2)"""
    from cofii2p_amd import dataside, synth as _synth

    raw, rimg, rK = _synth.make_raw_scan(0)
    cal = dataside.calib_matrices(_synth.KITTI_CALIB_LINES)
    P_Tr = np.dot(cal["P2"], cal["Tr"])
    opt_ds = Opt()
    for k_, v_ in dict(num_pc=points, num_kpt=64, P_tx_amplitude=10, P_ty_amplitude=0, P_tz_amplitude=10, P_Rx_amplitude=0.0, P_Ry_amplitude=2.0 * np.pi, P_Rz_amplitude=0.0).items():
        setattr(opt_ds, k_, v_)
    return opt_ds, raw, torch.from_numpy(raw).to(dev), torch.from_numpy(rimg).to(dev), rK, P_Tr


def loader_stack_pipeline(model, dev, opt_ds, st, Bsz, nsub, upk, raw_d, img_d, rK, P_Tr, slot_base, workers=4, KS=4):
    """The device-side loader feeding the headline's stack-mode submissions: stream s prepares the Bsz frames of a batch one after the other
    in its own KS loader slots (voxel grid of the next frames enqueued ahead), every finished frame is copied into its row block of a
    static stack (preprocess.FrameStack: one launch; the slot is free again at once), the forward of the whole batch follows on the same
    stream; the labels of a batch (kitti.py:333-420) are computed from one device-to-host copy of its coarsest points when its forward
    is collected.  2 S stacks in the ring.  -> (frames/s of the second pass, host seconds by call)."""
    from cofii2p_amd.loader import FrameLoader
    from cofii2p_amd.preprocess import FrameStack

    S = len(st)
    loader = FrameLoader(opt_ds, dev, slots=S * KS, workers=workers, capture_stream=st[0], upsample_k=upk)
    stacks, pend, ctxs = [None] * (2 * S), [None] * (2 * S), [[] for _ in range(2 * S)]
    host = {"begin": 0.0, "complete": 0.0, "put": 0.0, "forward": 0.0, "collect": 0.0}

    def timed(name, fn, *a, **k):
        t_ = time.perf_counter()
        r = fn(*a, **k)
        host[name] += time.perf_counter() - t_
        return r

    def collect(j):
        model.finish(pend[j])
        pyr = stacks[j].pyr
        n4, n1 = pyr["points"][-1].shape[0] // Bsz, pyr["points"][1].shape[0] // Bsz
        coarse = pyr["points"][-1].cpu().numpy()   # one copy for the labels of the whole batch
        for f, ctx in enumerate(ctxs[j]):
            loader.labels_from(coarse[f * n4:(f + 1) * n4], pyr["points"][1][f * n1:(f + 1) * n1], pyr["points"][-1][f * n4:(f + 1) * n4], ctx)
        ctxs[j] = []
        pend[j] = None

    try:
        for phase in range(2):
            for k_ in host:
                host[k_] = 0.0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # batch q is prepared frame after frame on stream q % S while the other streams execute the forwards of the batches before it: the
            # loader's chains of small kernels run under big stack-mode launches.  (Filling S stacks at the same time, a frame each in turn,
            # measured 374-384 frames/s against 422-459: four loader chains, then four forwards, each phase alone on the chip.)
            for q in range(nsub):
                s_, j = q % S, q % (2 * S)
                if pend[j] is not None:
                    timed("collect", collect, j)
                with torch.cuda.stream(st[s_]):
                    begun = 0
                    for f in range(Bsz):
                        while begun < min(Bsz, f + KS):   # the voxel grids of the next KS - 1 frames of this batch are already enqueued
                            timed("begin", loader.begin, s_ * KS + begun % KS, raw_d, img_d, rK, P_Tr, q * Bsz + begun)
                            begun += 1
                        loader.poll()
                        smp = timed("complete", loader.complete, s_ * KS + f % KS)
                        if stacks[j] is None:
                            stacks[j] = FrameStack(smp["pc_data_dict"], smp["pc_data_dict"]["feats"], smp["img"], Bsz)
                        timed("put", stacks[j].put, f, smp["pc_data_dict"], smp["pc_data_dict"]["feats"], smp["img"])
                        ctxs[j].append(smp["label_ctx"])
                        loader.release(s_ * KS + f % KS)
                    pend[j] = timed("forward", model.forward_async, slot_base + j, stacks[j].pyr, stacks[j].img, inputs_stable=True)
            for j in range(2 * S):
                if pend[j] is not None:
                    timed("collect", collect, j)
            torch.cuda.synchronize()
            dtl = time.perf_counter() - t0
            nframes = nsub * Bsz
    finally:
        loader.close()
    return nframes / dtl, {k_: v_ / nframes for k_, v_ in host.items()}


def loader_pipeline(model, dev, opt_ds, st, slots_per_stream, nfr, upk, raw_d, img_d, rK, P_Tr, slot_base, workers=4):
    """The pipelined loader (cofii2p_amd/loader.py) in front of the forward: voxel grid enqueued LOOK frames ahead, draws in worker
    processes, resample + pyramid + image as one hipGraph per slot, tables read in place by the forward's graph; INFL forwards in flight.
    Two passes (the first captures the graphs) -> (frames/s of the second, host seconds inside each call, INFL, LOOK)."""
    from cofii2p_amd.loader import FrameLoader

    LOOK, INFL = len(st), len(st) * max(1, slots_per_stream)
    NSL = INFL + LOOK
    loader = FrameLoader(opt_ds, dev, slots=NSL, workers=workers, capture_stream=st[0], upsample_k=upk)
    pend = [None] * NSL
    host = {"begin": 0.0, "complete": 0.0, "forward": 0.0, "collect": 0.0}   # host seconds inside each call (timed phase)

    def timed(name, fn, *a, **k):
        t_ = time.perf_counter()
        r = fn(*a, **k)
        host[name] += time.perf_counter() - t_
        return r

    def collect(sl):
        h, smp = pend[sl]
        model.finish(h)
        smp["finish_labels"]()
        loader.release(sl)
        pend[sl] = None

    try:
        for phase in range(2):
            for k_ in host:
                host[k_] = 0.0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(min(LOOK, nfr)):
                with torch.cuda.stream(st[j % len(st)]):
                    loader.begin(j % NSL, raw_d, img_d, rK, P_Tr, j)
            for i in range(nfr):
                sl, nxt = i % NSL, i + LOOK
                with torch.cuda.stream(st[i % len(st)]):
                    if nxt < nfr:   # the voxel grid of frame i + LOOK goes onto this stream AHEAD of frame i's own work
                        if pend[nxt % NSL] is not None:
                            timed("collect", collect, nxt % NSL)
                        timed("begin", loader.begin, nxt % NSL, raw_d, img_d, rK, P_Tr, nxt)
                    loader.poll()
                    smp = timed("complete", loader.complete, sl)
                    pend[sl] = (timed("forward", model.forward_async, slot_base + sl, smp["pc_data_dict"], smp["img"][None], inputs_stable=True), smp)
                loader.poll()
            for k in range(NSL):
                if pend[(nfr + k) % NSL] is not None:
                    timed("collect", collect, (nfr + k) % NSL)
            torch.cuda.synchronize()
            dtl = time.perf_counter() - t0
    finally:
        loader.close()
    return nfr / dtl, dict(host), INFL, LOOK


class optional_leg:
    """Everything after the headline measurement is additional information: a leg that fails (a worker pool that cannot spawn, a missing
    fixture, ...) is recorded in the line as `<name>_error` instead of costing the line."""

    def __init__(self, result, name):
        self.result, self.name = result, name

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception):
            self.result[self.name + "_error"] = "%s: %s" % (et.__name__, ev)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            return True
        return False


def timed_repeats(run_steps, barrier, steps, repeats):
    """`repeats` timed regions of EXACTLY `steps` steps each, every one bracketed by barrier + synchronize on both sides -> seconds per region"""
    out = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        run_steps(steps)
        barrier()
        out.append(time.perf_counter() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=20480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying the hipGraph")
    ap.add_argument("--inflight", type=int, default=4, help="frame streams per GPU (HIP streams; each carries --slots-per-stream hipGraph slots)")
    ap.add_argument("--slots-per-stream", type=int, default=2, help="submissions queued per frame stream: 2 = the next frame is already enqueued behind the running one (no host bubble)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for single-GPU tests of the N > 1 path)")
    ap.add_argument("--share-device", action="store_true", help="test aid: all ranks use cuda:0")
    ap.add_argument("--copy-inputs", action="store_true", help="stage every frame's inputs into per-slot static buffers (one 27 MB copy launch per frame) "
                    "instead of letting the hipGraph read the resident input tensors in place (forward_async(inputs_stable=True))")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the extra stack-mode batch-4/16 measurements")
    ap.add_argument("--batch", type=int, default=16, help="frames per submission (stack mode; BASELINE configs[2] = 16, configs[1] = 1); a step is one submission")
    ap.add_argument("--stress", action="store_true", help="bench BASELINE configs[4] instead: 896x1600 image, 40960 points (implies --points 40960)")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps each; the median is reported (0 = 5 when --steps < 100, else 1)")
    ap.add_argument("--distinct-frames", type=int, default=16, help="distinct synthetic frames per rank cycled by the timed loop (16 x 27 MB of "
                    "tables do not fit the 256 MB Infinity Cache)")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra measurements in the other arithmetics (other_arithmetics)")
    ap.add_argument("--loader-leg", action="store_true", help="N > 1: every rank also runs the device-side loader pipeline (row f2) next to the others")
    ap.add_argument("--no-steady", action="store_true", help="skip the long-region re-measurement of the headline loop (steady_state)")
    ap.add_argument("--gemm", default=os.environ.get("COFI_GEMM", "bf16x6"), choices=["f32", "bf16x3", "bf16x6"],
                    help="arithmetic of the dense contractions: exact fp32 MFMA, 3-term bf16 split, or 6-term bf16 split (fp32-grade), all with fp32 accumulation")
    args = ap.parse_args()
    if args.stress:
        Opt.img_H, Opt.img_W, args.points = 896, 1600, 40960

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become N ranks (one process per GPU) under torch.distributed.run
        import socket
        import subprocess

        have = torch.cuda.device_count()
        if have < args.gpus and not args.share_device:
            raise SystemExit("--gpus %d but this node exposes %d GPU(s)" % (args.gpus, have))
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.share_device:  # test aid: every rank on cuda:0 (with --dist-backend gloo; RCCL refuses two ranks on one GPU)
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d wants cuda:%d but this node exposes %d GPU(s)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)   # BEFORE the process group exists: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this host driver only supports dmabuf IPC (RCCL / xGMI peer access)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
    numa_node = pin_to_gpu_numa_node(local) if (world > 1 and not args.share_device) else None

    from cofii2p_amd import ops as cofi_ops
    from cofii2p_amd.network import CoFiI2P
    from cofii2p_amd.parallel import shard_frames

    cofi_ops.GEMM_MODE = args.gemm
    CoFiI2P.MAX_STABLE_GRAPHS = 512   # this process measures the same loops in three arithmetics and several pipelines: one graph per (slot, input set, arithmetic)
    model = CoFiI2P(Opt()).to(dev)
    _model_ref.append(model)
    if not args.eager:
        model.enable_graphs()
    n_distinct = max(1, args.distinct_frames)
    my_ids = shard_frames(list(range(n_distinct * world)), rank, world)
    frames = make_inputs(dev, my_ids, args.points)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    nmatch = 0
    S = max(1, args.inflight) if not args.eager else 1
    Bsz = max(1, args.batch)
    repeats = args.repeats if args.repeats > 0 else (5 if args.steps < 100 else 1)
    pipe = bpipe = None
    if Bsz > 1:
        # stack mode: Bsz frames per submission through the same launches, S submissions in flight
        bpipe = BatchPipeline(model, dev, frames, Bsz, S, args.copy_inputs)
        nmatch = bpipe.warm(args.warmup)
        dts = timed_repeats(bpipe.run, barrier, args.steps, repeats)
    elif S == 1:
        def run1(nsteps):
            for i in range(nsteps):
                one_step(model, frames[i % len(frames)])

        for i in range(max(args.warmup, len(frames))):
            out, _ = one_step(model, frames[i % len(frames)])
            nmatch = out[4].shape[0]
        dts = timed_repeats(run1, barrier, args.steps, repeats)
    else:
        pipe = Pipeline(model, dev, frames, S, args.slots_per_stream, args.copy_inputs)
        nmatch = pipe.warm(args.warmup)
        dts = timed_repeats(pipe.run, barrier, args.steps, repeats)
    headline = bpipe if bpipe is not None else pipe
    dt = float(np.median(dts))   # this rank's seconds per timed region of args.steps steps
    peak_headline_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30   # weights + planes + inputs + one private pool per captured hipGraph
    gathered = per_rank = gather_ms = None
    rccl = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        dt = max(per_rank)   # the job is as fast as its slowest rank
        # the one exchange of a frame-parallel evaluation: per-frame results of every rank -> all ranks, global
        # frame order, over the process group the ranks were timed in (RCCL over xGMI with the default backend)
        from cofii2p_amd.parallel import gather_frame_results

        vals = torch.tensor([[float(nmatch), float(rank)] for _ in my_ids], dtype=torch.float32, device=dev if args.dist_backend == "nccl" else "cpu")
        barrier()
        t0 = time.perf_counter()
        gathered = gather_frame_results(my_ids, vals.reshape(len(my_ids), 2), n_distinct * world)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - t0)
        assert sorted(set(int(r) for r in gathered[:, 1].tolist())) == list(range(world))
        # proof of the ranks for the driver's scaling run: the ranks the process group reports, the distinct devices they sit on (an
        # all_gather of every rank's PCI bus id), the collective library's version
        bus = torch.tensor([torch.cuda.get_device_properties(local).pci_bus_id, local, rank], dtype=torch.int64,
                           device=dev if args.dist_backend == "nccl" else "cpu")
        allb = [torch.zeros_like(bus) for _ in range(world)]
        dist.all_gather(allb, bus)
        rccl = {"backend": args.dist_backend, "ranks": dist.get_world_size(), "rank_devices": [[int(v) for v in b.tolist()] for b in allb],
                "distinct_devices": len({int(b[0]) for b in allb}) if not args.share_device else 1,
                "nccl_version": _nccl_version() if args.dist_backend == "nccl" else None}
    fps = world * args.steps * Bsz / dt
    arith_note = {"f32": "on the exact fp32 MFMA (bit-equal to an fmaf chain)",
                  "bf16x3": "as a 3-term bf16 split (hi*hi + hi*lo + lo*hi) on the bf16 matrix cores: ~2^-16 per product, 2e-5 max abs deviation from the "
                            "reference's outputs on the golden frame (budget 1e-3) - narrower than the reference's fp32",
                  "bf16x6": "as a 6-term bf16 split (hi/mid/lo planes of both operands = all 24 mantissa bits, six products on v_mfma_f32_32x32x16_bf16): "
                            "fp32-GRADE - the same error against fp64 as the exact-fp32 kernel (2e-6 on the golden frame); the arithmetic the "
                            "reference-named class model.network.CoFiI2P ships"}[args.gemm]
    result = {
        "metric": "I2P frames/sec (160x512 img, 20480 pts)", "value": fps, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        # the arithmetic the dense contractions compute in; storage, accumulation and KPConv aggregation are fp32 in all of them, the
        # attention kernel runs in the same 6-term split (bf16x6, bf16x3 runs) or on the exact fp32 instruction (f32 runs): "attention_arithmetic"
        "dtype": args.gemm,
        "repeats": repeats, "seconds_per_repeat": [round(x, 6) for x in dts], "distinct_frames_per_rank": len(frames),
        "frames_per_step": Bsz, "ms_per_frame": 1e3 * dt / (args.steps * Bsz),
        "peak_mem_GB": round(peak_headline_gb, 2),
        "data": "synthetic", "launch": "eager" if args.eager else "hipGraph replay",
        "input_staging": "copied into per-slot static buffers" if (args.copy_inputs or args.eager) else "read in place (inputs resident in HBM, forward_async(inputs_stable=True))", "gemm_mode": args.gemm,
        "ranks": world if dist is None else dist.get_world_size(), "dist_backend": None if dist is None else args.dist_backend, "rccl": rccl,
        "gathered_frame_results": None if gathered is None else int(gathered.shape[0]),
        "per_rank_frames_per_s": None if per_rank is None else [round(args.steps * Bsz / x, 2) for x in per_rank],
        "result_gather_ms": gather_ms, "numa_node": numa_node,
        "arithmetic": "fp32 storage and accumulation everywhere; dense contractions " + arith_note,
        "attention_arithmetic": __import__("cofii2p_amd.ops", fromlist=["ops"]).attention_arith(),
        # north star / SURVEY.md 8(d): throughput as a fraction of the attention roofline = frames/s x 13.42 GFLOP of attention per
        # frame (4 self + 4 cross layers over both token streams: 4 * 512 * (1280 + 1280)^2) / the matrix-pipe roof of the GPUs used in the
        # attention kernel's arithmetic (attention_peak_tf: 416.7 TF/s for the bf16x6 kernel, 157.3 for the fp32-instruction kernel)
        "attention_roofline_frac": fps * 4 * 512 * ((Opt.img_H // 8) * (Opt.img_W // 8) + args.points // 16) ** 2 / (world * attention_peak_tf() * 1e12),
        "config": {"workload": "%s synthetic frames (%dx%d image, %d points, KNN-128 pyramid resident in HBM), CoFiI2P.forward(mode='test') + fine "
                               "matching, %s per step per GPU, %d submissions in flight%s"
                               % ("KITTI-shape" if not args.stress else "stress (BASELINE configs[4])", Opt.img_H, Opt.img_W, args.points,
                                  "one frame (batch 1, BASELINE configs[1])" if Bsz == 1 else
                                  "one stack-mode batch of %d frames%s" % (Bsz, " (BASELINE configs[2])" if Bsz == 16 else ""), S,
                                  "" if Bsz == 1 else "; the batch-1 pipeline (configs[1]) is in config.batch1_frames_per_s"),
                   "batch": Bsz, "arithmetic": args.gemm,
                   "matches_per_frame": int(nmatch), "parallelism": "frame-parallel x%d" % world, "frame_streams_per_gpu": S,
                   "hipgraph_slots_per_stream": max(1, args.slots_per_stream) if (S > 1 and Bsz == 1) else 1},
    }
    if world > 1 and args.loader_leg and not args.eager:
        # multi-GPU readiness: every rank runs the device-side loader (its own FrameLoader worker pool, its NUMA pinning)
        # in front of its forwards at the same time as the other ranks; the per-rank rates are gathered over the process group
        leg_err = None
        try:
            opt_ds, _raw, raw_d, img_d, rK, P_Tr = dataside_inputs(dev, args.points)
            st = make_streams(dev, max(S, 1))
            model.enable_graphs(True)
            rate = loader_pipeline(model, dev, opt_ds, st, args.slots_per_stream, max(8, 2 * args.steps), 1, raw_d, img_d, rK, P_Tr, 100, workers=2)[0]
        except Exception as e:   # noqa: BLE001 - an extra leg: reported, never fatal
            rate, leg_err = 0.0, "%s: %s" % (type(e).__name__, e)
        rates = [None] * world
        dist.all_gather_object(rates, (rate, leg_err))
        result["loader_leg"] = {"per_rank_frames_per_s": [r[0] for r in rates], "errors": [r[1] for r in rates if r[1]],
                                "note": "FrameLoader (voxel grid + resample + KNN pyramid + image, nearest-only up-sampling tables) + forward + fine matching, "
                                        "pipelined, on every rank concurrently"}
    extras = rank == 0 and world == 1 and not args.eager and not args.no_batch_sweep
    if rank == 0 and world == 1 and not args.eager and S > 1 and not args.no_steady:
        with optional_leg(result, "steady_state"):
            # the driver's 20 steps put a pipeline fill and drain inside a short timed region: the same loop over a longer one, on record
            nst = max(args.steps, (1600 + Bsz - 1) // Bsz if Bsz > 1 else 400)
            d_long = float(np.median(timed_repeats(headline.run, barrier, nst, 1)))
            result["steady_state"] = {"steps": nst, "frames_per_s": nst * Bsz / d_long, "seconds": round(d_long, 4),
                                      "note": "identical loop, one long timed region: `value` (%d steps) carries the fill / drain of the %d-deep pipeline" % (args.steps, S)}
            result["config"]["steps_for_steady_state"] = nst
            result["config"]["steady_state_frames_per_s"] = nst * Bsz / d_long
    if extras and Bsz != 1 and S > 1:
        with optional_leg(result, "batch1_pipeline"):
            # BASELINE configs[1]: one frame per submission, S frame streams x slots_per_stream hipGraph slots (the round 1-3 headline pipeline)
            pipe = Pipeline(model, dev, frames, S, args.slots_per_stream, args.copy_inputs)
            pipe.warm(args.warmup)
            n1 = max(args.steps, 200)
            d1 = float(np.median(timed_repeats(pipe.run, barrier, n1, 3)))
            result["batch1_pipeline"] = {"frames_per_s": n1 / d1, "ms_per_frame": 1e3 * d1 / n1, "frames": n1, "arithmetic": args.gemm, "frames_in_flight": S * max(1, args.slots_per_stream),
                                         "note": "BASELINE configs[1]: ONE frame per submission (batch 1), %d frame streams x %d hipGraph slots" % (S, max(1, args.slots_per_stream))}
            result["config"]["batch1_frames_per_s"] = n1 / d1
    if extras and Bsz > 1:
        with optional_leg(result, "single_frame_api"):
            # the same frames handed over ONE AT A TIME (cofii2p_amd.serving.FrameBatcher.submit / result): every frame is copied into the
            # stack being filled (one launch, 27 MB) and the stack runs as a stack-mode submission of Bsz frames - the batch rate behind a
            # single-frame API
            from cofii2p_amd.serving import FrameBatcher

            fb = FrameBatcher(model, batch=Bsz, streams=S, slot_base=220)
            nfr = max(args.steps, 40) * Bsz
            for phase in range(2):   # 0: captures the graphs of the ring's slots
                tickets = []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(nfr if phase else 2 * S * Bsz):
                    pyr_i, img_i, _ = frames[i % len(frames)]
                    tickets.append(fb.submit(pyr_i, img_i))
                    if len(tickets) > (2 * S - 1) * Bsz:        # read results one ring behind the submissions
                        fb.result(tickets.pop(0))
                while tickets:
                    fb.result(tickets.pop(0))
                torch.cuda.synchronize()
                d_api = time.perf_counter() - t0
            result["single_frame_api"] = {"frames_per_s": nfr / d_api, "ms_per_frame": 1e3 * d_api / nfr, "frames": nfr, "batch": Bsz,
                                          "note": "serving.FrameBatcher: submit(frame) / result(ticket) per frame, executed as stack-mode submissions of %d frames "
                                                  "(each frame copied into its stack by one launch)" % Bsz}
            result["config"]["single_frame_api_frames_per_s"] = nfr / d_api
    if extras and not args.no_f32:
        with optional_leg(result, "other_arithmetics"):
            # the same loops in the other two arithmetics, on record next to `value`: the narrower 3-term split (faster) and the exact fp32 MFMA
            oth = {}
            for mode in ("bf16x3", "bf16x6", "f32"):
                if mode == args.gemm:
                    continue
                cofi_ops.GEMM_MODE = mode
                try:
                    headline.warm(args.warmup)
                    dh = float(np.median(timed_repeats(headline.run, barrier, args.steps, repeats)))
                    ent = {"frames_per_s": args.steps * Bsz / dh}
                    if pipe is not None and headline is not pipe:
                        pipe.warm(args.warmup)
                        db = float(np.median(timed_repeats(pipe.run, barrier, max(args.steps, 100), 1)))
                        ent["batch1_frames_per_s"] = max(args.steps, 100) / db
                finally:
                    cofi_ops.GEMM_MODE = args.gemm
                oth[mode] = ent
                result["config"]["frames_per_s_" + mode] = ent["frames_per_s"]
            result["other_arithmetics"] = oth
            result["other_arithmetics"]["note"] = ("identical loops with COFI_GEMM=<mode>: bf16x3 = 3-term split (~2^-16 per product, narrower than fp32), "
                                                   "f32 = exact fp32 MFMA, bf16x6 = fp32-grade 6-term split")
    if extras:
        with optional_leg(result, "forward_sync"):
            # the reference-surface call pattern (evaluation/eval_all.py:94-96): model(...) per frame, one host synchronisation per
            # frame, nothing in flight behind it - what a caller gets without forward_async / finish
            n_sync = max(20, min(100, args.steps))
            for i in range(4):
                one_step(model, frames[i % len(frames)])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_sync):
                one_step(model, frames[i % len(frames)])
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
            # ... and what a caller that only swaps the import gets: the reference-named class with ITS defaults (fp32-grade bf16x6
            # arithmetic, hipGraph replay unasked with results the caller owns)
            from model.network import CoFiI2P as ShimCoFiI2P

            shim = ShimCoFiI2P(Opt()).to(dev)
            with torch.no_grad():
                for i in range(4):
                    one_step(shim, frames[i % len(frames)])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n_sync):
                    one_step(shim, frames[i % len(frames)])
                torch.cuda.synchronize()
            dshim = time.perf_counter() - t0
            del shim
            result["forward_sync"] = {"frames_per_s": n_sync / dts, "ms_per_frame": 1e3 * dts / n_sync, "frames": n_sync,
                                      "reference_named_class_defaults": {"frames_per_s": n_sync / dshim, "arithmetic": "bf16x6",
                                                                         "note": "`from model.network import CoFiI2P`, nothing else changed"},
                                      "note": "model(pc_data_dict, img, ..., 'test') per frame as eval_all.py:94-96 calls it (hipGraph replay, one host "
                                              "sync per frame, no frames in flight); `value` is the pipelined forward_async / finish rate"}
    if rank == 0 and not args.no_kernel_timing:
        with optional_leg(result, "roofline"):
            model.enable_graphs(False)
            if Bsz == 1:
                result.update(kernel_rooflines(model, dev, args, 1, frame=frames[0]))
            else:
                result.update(kernel_rooflines(model, dev, args, Bsz, batch=bpipe.batches[0]))
                if extras:   # ... and the same rows for one frame per submission (BASELINE configs[1])
                    result["batch1_rooflines"] = kernel_rooflines(model, dev, args, 1, frame=frames[0])
            # GPU dispatches per submission, counted in the captured hipGraph (kernel + copy + memset nodes): `launches_per_frame` above
            # counts C-ABI entry-point calls, several of which launch more than one kernel
            with optional_leg(result, "dispatch_census"):
                cen = {"one_frame": submission_census(model, dev, [frames[0]])}
                if Bsz > 1:
                    cen["batch_of_%d" % Bsz] = submission_census(model, dev, list(frames[:Bsz]) if len(frames) >= Bsz else [frames[i % len(frames)] for i in range(Bsz)])
                cen["note"] = "nodes of the hipGraph one forward(mode='test') + fine matching is captured into (hipGraphGetNodes), by node type"
                result["dispatch_census"] = cen
                if cen["one_frame"]:
                    result["config"]["batch1_dispatches_per_frame"] = cen["one_frame"]["dispatches"]
            rf = result.get("roofline", {})
            if rf.get("bound") == "mfma" and "algorithmic_gflop_per_frame" in rf:
                # `frac` prices one launch at a time (isolated replay); with submissions in flight the kernels share the chip, so the family's
                # share of the chip over the whole timed region is its work per frame x frames/s (per GPU) over the same peak
                rf["chip_level_frac"] = rf["algorithmic_gflop_per_frame"] * 1e9 * (result["value"] / world) / (rf["peak"] * 1e12)
            rc = result.get("roofline_cross_attention")
            if rc:   # the north star's second number, where the driver's parser keeps it
                result["config"]["cross_attention_mfma_frac"] = rc["frac"]
    if extras:
        with optional_leg(result, "stack_mode_batches"):
            # additional information: the same frames in stack-mode batches of other sizes through the same kernels
            model.enable_graphs(True)
            sweep = {}
            for bsz in [b for b in (4, 8, 16) if b != Bsz]:
                grp = [frames[i % len(frames)] for i in range(bsz)]
                pyr_b, img_b = CoFiI2P.stack_frames([g[0] for g in grp], [g[1] for g in grp])
                st = make_streams(dev, S)
                pend = [None] * S
                nst = max(2 * S, args.steps // bsz)
                for phase in range(2):  # 0 = warm-up (captures the graphs), 1 = timed
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(nst):
                        sl = i % S
                        if pend[sl] is not None:
                            model.finish(pend[sl])
                        with torch.cuda.stream(st[sl]):
                            pend[sl] = model.forward_async(10 + sl, pyr_b, img_b)
                    for sl in range(S):
                        if pend[sl] is not None:
                            model.finish(pend[sl])
                            pend[sl] = None
                    torch.cuda.synchronize()
                    dtb = time.perf_counter() - t0
                sweep["batch_%d" % bsz] = {"frames_per_s": nst * bsz / dtb, "ms_per_frame": 1e3 * dtb / (nst * bsz), "submissions_in_flight": S}
                if bsz == 16 and Bsz == 1 and not args.no_kernel_timing:   # BASELINE configs[2]: the same roofline rows for the stacked batch
                    model.enable_graphs(False)
                    sweep["batch_16"].update(kernel_rooflines(model, dev, args, bsz, batch=(pyr_b, img_b)))
                    model.enable_graphs(True)
                del pyr_b, img_b
            result["stack_mode_batches"] = sweep
    if rank == 0 and world == 1 and not args.no_batch_sweep:
        with optional_leg(result, "knn_pyramid"):
            # additional information, outside `value` (the reference's DataLoader builds the pyramid, preprocess_data.py:36-107): the 13
            # KNN-128 searches of one frame's pyramid on this GPU, cell-grid search vs the brute-force kernel (identical tables)
            from cofii2p_amd import ops as _ops
            from cofii2p_amd.preprocess import build_pyramid
            from cofii2p_amd.synth import subsample_indices

            p0 = frames[0][0]["points"][0]
            sub = [torch.from_numpy(s_).to(dev) for s_ in subsample_indices(args.points, 5, seed=1000)]

            def pyramid_ms(n=20):
                for _ in range(3):
                    build_pyramid(p0, sub)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    build_pyramid(p0, sub)
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t0) / n

            saved = _ops.KNN_GRID_MIN_SUPPORT
            grid_ms = pyramid_ms()
            _ops.KNN_GRID_MIN_SUPPORT = 1 << 30
            brute_ms = pyramid_ms()
            _ops.KNN_GRID_MIN_SUPPORT = saved
            result["knn_pyramid"] = {"ms_per_frame": grid_ms, "brute_force_ms_per_frame": brute_ms,
                                     "note": "build_pyramid (5 stages, 13 searches, k = 128) as called from Python, one stream; not part of `value`"}
            if S > 1 and not args.eager:
                # ... and the whole chain on this GPU: every frame's pyramid is built on its frame stream right before its forward - the ~40
                # launches of a pyramid as ONE hipGraph per slot (preprocess.PyramidGraph), its tables read in place by the forward's graph
                from cofii2p_amd.preprocess import PyramidGraph

                model.enable_graphs(True)
                st = make_streams(dev, S)
                NSL = S * max(1, args.slots_per_stream)
                pend = [None] * NSL
                feats0 = frames[0][0]["feats"]
                sub_sizes = [int(t.shape[0]) for t in sub]
                rates = {}
                for upk in (None, 1):   # the reference's (N, 128) up-sampling tables / their first column only (all the forward reads), derived without a search
                    pgs = [PyramidGraph(args.points, sub_sizes, dev, capture_stream=st[0], upsample_k=upk) for _ in range(NSL)]
                    imgs = [frames[0][1].clone() for _ in range(NSL)]   # static per slot, like the tables
                    for phase in range(2):   # 0 = warm-up (captures the graphs of these slots), 1 = timed
                        nfr = max(args.steps, 3 * NSL)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for i in range(nfr):
                            sl = i % NSL
                            if pend[sl] is not None:
                                model.finish(pend[sl])
                            with torch.cuda.stream(st[i % S]):
                                pyr = dict(pgs[sl].run(p0, sub))
                                pyr["feats"] = feats0
                                pend[sl] = model.forward_async((30 if upk is None else 45) + sl, pyr, imgs[sl], inputs_stable=True)
                        for sl in range(NSL):
                            if pend[sl] is not None:
                                model.finish(pend[sl])
                                pend[sl] = None
                        torch.cuda.synchronize()
                        dte = time.perf_counter() - t0
                    rates[upk] = (nfr / dte, 1e3 * dte / nfr)
                    del pgs
                stacked = None
                if Bsz > 1:
                    # ... and the same chain feeding the headline's stack-mode submissions: S PyramidGraphs (one per stream) build the frames of a
                    # batch one after the other, each is copied into its row block of a static stack (preprocess.FrameStack, one launch), the
                    # forward of the whole batch follows on the same stream; 2 S stacks in the ring
                    from cofii2p_amd.preprocess import FrameStack

                    stacked = {}
                    img0 = frames[0][1]
                    for upk in (None, 1):
                        pgs = [PyramidGraph(args.points, sub_sizes, dev, capture_stream=st[0], upsample_k=upk) for _ in range(S)]
                        tmpl = pgs[0].run(p0, sub)
                        stacks = [FrameStack(tmpl, feats0, img0, Bsz) for _ in range(2 * S)]
                        pendb = [None] * (2 * S)
                        nsub = max(4 * S, args.steps)
                        for phase in range(2):
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            for i in range(nsub):
                                j = i % (2 * S)
                                if pendb[j] is not None:
                                    model.finish(pendb[j])
                                with torch.cuda.stream(st[i % S]):
                                    for f in range(Bsz):
                                        stacks[j].put(f, pgs[i % S].run(p0, sub), feats0, img0)
                                    pendb[j] = model.forward_async((120 if upk is None else 140) + j, stacks[j].pyr, stacks[j].img, inputs_stable=True)
                            for j in range(2 * S):
                                if pendb[j] is not None:
                                    model.finish(pendb[j])
                                    pendb[j] = None
                            torch.cuda.synchronize()
                            dts_ = time.perf_counter() - t0
                        stacked["reference_tables" if upk is None else "nearest_only_upsampling"] = {"frames_per_s": nsub * Bsz / dts_, "ms_per_frame": 1e3 * dts_ / (nsub * Bsz)}
                        del pgs, stacks
                    stacked["note"] = ("every frame's pyramid built on the GPU (one hipGraph per frame) and copied into its row block of a static stack "
                                       "(preprocess.FrameStack), then ONE stack-mode forward of %d frames: the headline's submissions with the pyramid in the chain" % Bsz)
                    result["config"]["with_pyramid_build_frames_per_s"] = stacked["nearest_only_upsampling"]["frames_per_s"]
                result["with_pyramid_build"] = {"frames_per_s": rates[None][0], "ms_per_frame": rates[None][1], "stack_mode_submissions": stacked,
                                                "nearest_only_upsampling": {"frames_per_s": rates[1][0], "ms_per_frame": rates[1][1],
                                                                            "note": "build_pyramid(upsample_k=1): the four up-sampling tables hold their first column only - "
                                                                                    "all the forward reads (functional.py:20) - derived from neighbors[i] without a search; "
                                                                                    "outputs bit-identical"},
                                                "note": "pyramid construction (5 cell grids + 9 KNN-128 searches + 4 row gathers, one hipGraph) + forward + fine matching per "
                                                        "frame on the same GPU; not the headline"}
    if extras:
        with optional_leg(result, "with_dataside"):
            # row f2: the whole data side of a frame on this GPU (data/kitti.py:259-393: calibration transform, 0.1 m voxel grid, resample to
            # num_pc, random SE(3), KNN pyramid, image resize / crop, labels) from a raw 120 000-point scan + 376 x 1241 image already in
            # HBM, alone and in front of the forward
            from cofii2p_amd import dataside

            opt_ds, raw, raw_d, img_d, rK, P_Tr = dataside_inputs(dev, args.points)
            st = make_streams(dev, max(S, 1))
            prep0 = dataside.FramePreparer(opt_ds, dev)
            for i in range(3):
                prep0.prepare(raw_d, img_d, rK, P_Tr, i)
            torch.cuda.synchronize()
            nl = 20
            t0 = time.perf_counter()
            for i in range(nl):
                prep0.prepare(raw_d, img_d, rK, P_Tr, i)
            torch.cuda.synchronize()
            loader_ms = 1e3 * (time.perf_counter() - t0) / nl
            voxels = prep0.last["voxels"]
            del prep0
            model.enable_graphs(True)
            ds_rates = {}
            nfr = max(args.steps, 3 * (len(st) * (max(1, args.slots_per_stream) + 1)))
            for upk in (None, 1):   # the reference's (N, 128) up-sampling tables / nearest-only tables derived without a search (outputs bit-identical)
                rate, hst, INFL, LOOK = loader_pipeline(model, dev, opt_ds, st, args.slots_per_stream, nfr, upk, raw_d, img_d, rK, P_Tr, 60 if upk is None else 80)
                ds_rates[upk] = (rate, hst)
            stacked_ds = None
            if Bsz > 1:
                with optional_leg(result, "with_dataside_stack"):
                    stacked_ds = {}
                    for upk in (None, 1):
                        r_, h_ = loader_stack_pipeline(model, dev, opt_ds, st, Bsz, max(3 * len(st), args.steps // 2), upk, raw_d, img_d, rK, P_Tr, 160 if upk is None else 180)
                        stacked_ds["reference_tables" if upk is None else "nearest_only_upsampling"] = {
                            "frames_per_s": r_, "host_ms_per_frame_in": {k_: round(1e3 * v_, 4) for k_, v_ in h_.items()}}
                    stacked_ds["note"] = ("the loader feeding the headline's stack-mode submissions of %d frames (preprocess.FrameStack; labels of a batch computed "
                                          "when its forward is collected)" % Bsz)
                    # the figure quoted next to `value` is the one with the REFERENCE's tables (all 128 upsampling columns, preprocess_data.py:55-99)
                    result["config"]["with_dataside_frames_per_s"] = stacked_ds["reference_tables"]["frames_per_s"]
                    result["config"]["with_dataside_nearest_only_frames_per_s"] = stacked_ds["nearest_only_upsampling"]["frames_per_s"]
            dtl, host = nfr / ds_rates[None][0], ds_rates[None][1]
            result["with_dataside"] = {"loader_ms_per_frame": loader_ms, "loader_frames_per_s": 1e3 / loader_ms, "frames_per_s": nfr / dtl,
                                       "ms_per_frame": 1e3 * dtl / nfr, "voxels": voxels, "raw_points": int(raw.shape[1]),
                                       "frames_in_flight": INFL, "voxel_grids_ahead": LOOK, "draw_workers_requested": 4,
                                       "draw_workers": __import__("cofii2p_amd.loader", fromlist=["FrameLoader"]).FrameLoader.worker_budget(4),   # what FrameLoader actually starts (affinity mask / ranks per node)
                                       "nearest_only_upsampling_frames_per_s": ds_rates[1][0], "stack_mode_submissions": stacked_ds,
                                       "host_ms_per_frame_in": {k_: round(1e3 * v_ / nfr, 4) for k_, v_ in host.items()},
                                       "note": "kitti.py:259-393 on the device (voxel grid + resample + SE(3) + KNN pyramid + image + labels) in front of the "
                                               "forward + fine matching, pipelined (cofii2p_amd/loader.py): raw scan and image resident in HBM, voxel count "
                                               "read asynchronously, Mersenne-Twister draws in worker processes, labels finished when the frame's forward is "
                                               "collected; loader_ms_per_frame = the synchronous FramePreparer.prepare() alone; not the headline"}
    if extras and not args.stress:
        with optional_leg(result, "stress_config"):
            result["stress_config"] = stress_summary(dev, args)
    if rank == 0 and world == 1 and not args.no_batch_sweep and not args.stress:
        # row f3, outside `value`: one optimisation step of train.py:186-286 on the same frame (forward(mode='train') -> the three losses ->
        # backward -> Adam), fp32-grade contractions (bf16x6); a separate module instance so the served one keeps its weights
        try:
            result["train_step"] = train_step_summary(dev, frames[0])
        except Exception as e:   # additional information only: never costs the line
            result["train_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with optional_leg(result, "cpu_baseline"):
            result["cpu_baseline"] = cpu_baseline(frames[0])
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
