"""Pipelined device-side loader (data/kitti.py:259-393 + preprocess_data.py:36-107): the stages of
`FramePreparer.prepare` spread over time so that neither the host thread nor the GPU waits for the other.

    begin(slot)     calibration transform + voxel grid of a frame, enqueued SEVERAL frames ahead on the frame's own stream; the voxel
                    count travels to a pinned host word, nobody waits for it
    poll()          slots whose count has arrived hand their random draws (Mersenne-Twister streams in the reference's call order:
                    ~1 ms of pure host work per frame, cofii2p_amd/sampler.py) to a pool of spawned worker PROCESSES (the draws hold
                    the GIL: threads do not scale)
    complete(slot)  the finished draws are copied into the slot's static device buffers and ONE hipGraph replays resample + SE(3) +
                    the 13-search KNN pyramid + image resize / crop (captured once per slot and image size); its outputs are static
                    tensors, int32 tables, which `CoFiI2P.forward_async(slot, ..., inputs_stable=True)` reads in place
    labels          the coarsest-stage points come back through a pinned buffer written by the same graph; the numpy label
                    projection runs when the caller asks for it (`sample["finish_labels"]()`), typically when the forward is collected
    mode='train'    kitti.py:312-314, 329-330: the frame's random crop (drawn on the main process from the generators the worker left behind)
                    and colour jitter; the image is then produced behind the replay (its crop is a kernel argument), a fresh tensor per frame

No extra HIP stream is created: HIP serves a process's streams from 4 hardware queues, and a fifth queue user costs the forward pipeline
more than the loader gains (DESIGN.md section 3 "Concurrency").  Stage A of a later frame is enqueued on its stream BEFORE the
resample / forward work of the frame the host is about to submit there, so its count is known long before it is needed."""
import multiprocessing as mp
import os
from concurrent.futures import ProcessPoolExecutor
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, dataside, ops
from .preprocess import build_pyramid
from .sampler import NUM_STAGES, draw_frame, sampler_from_state, worker_init, worker_warm


class _Slot:
    def __init__(self):
        self.phase = "idle"   # idle -> voxel (begin) -> drawing (job in the pool) -> ready (complete returned)
        self.generation = 0   # bumped by release(): a sample's lazy finish_labels() refuses to read the slot's buffers after that
        self.h = self.future = None
        self.graphs = {}      # image shape -> (graph, outputs)
        self.stage = None     # pinned staging + static device buffers of the draws


class FrameLoader:
    def __init__(self, opt, device="cuda", slots: int = 8, workers: int = 4, dataset: str = "kitti", capture_stream: Optional[torch.cuda.Stream] = None,
                 upsample_k: Optional[int] = None, mode: str = "val"):
        """slots: frames that can be in preparation / flight at once (a slot is reusable once the forward that read its tensors has been
        collected).  capture_stream: the stream hipGraphs are captured on (pass the model's: `model.frame_streams(n)[0]`); capture
        needs a non-default stream and every extra stream costs a hardware queue."""
        self.opt, self.device, self.dataset, self.mode = opt, torch.device(device), dataset, mode
        # upsample_k=1: the up-sampling tables hold their first column only (all the forward reads), derived without a search
        # (preprocess.build_pyramid); None: the reference's (N, 128) tables
        self.upsample_k = upsample_k
        self.preps = [dataside.FramePreparer(opt, device, dataset=dataset, mode=mode) for _ in range(slots)]
        self.slots = [_Slot() for _ in range(slots)]
        workers = self.worker_budget(workers)
        self.workers = workers
        self.pool = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"), initializer=worker_init)
        self._amp = tuple(getattr(opt, k) for k in ("P_tx_amplitude", "P_ty_amplitude", "P_tz_amplitude", "P_Rx_amplitude", "P_Ry_amplitude", "P_Rz_amplitude"))
        self._capture_stream = capture_stream
        # start every worker now (spawn + imports take seconds), not inside the first timed frames: W tasks that each hold a worker
        list(self.pool.map(worker_warm, [0.3] * workers))

    @staticmethod
    def worker_budget(requested: int, affinity=None, local_world: Optional[int] = None, online: Optional[int] = None) -> int:
        """Draw workers this process may start: the host cores its rank can count on, minus one for the launch thread.  One process per GPU
        (bench.py --gpus N, parallel.py): N ranks x `requested` workers must not oversubscribe the node - the cores are the smaller of this
        process's affinity mask (`os.sched_getaffinity`, e.g. the GPU's NUMA node after bench.pin_to_gpu_numa_node) and an equal share of
        the online cores among the LOCAL_WORLD_SIZE ranks of the node."""
        aff = len(affinity) if affinity is not None else len(os.sched_getaffinity(0))
        # LOCAL_WORLD_SIZE (torchrun / bench.py set it) = ranks on THIS node; without it: one rank per visible GPU at most - never the
        # global WORLD_SIZE, which on a multi-node launch would divide this node's cores by the ranks of every node
        lw = local_world if local_world is not None else int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
        if lw <= 0:
            lw = 1
        on = online if online is not None else (os.cpu_count() or aff)
        share = max(1, min(aff, on // max(1, lw)))
        return max(1, min(int(requested), share - 1))

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)

    # ------------------------------------------------------------------ stage A
    def begin(self, slot: int, data, img, K: np.ndarray, P_Tr: np.ndarray, index: int):
        """enqueue the calibration transform + voxel grid of frame `index` on the CURRENT stream into slot `slot` (no host wait)"""
        st = self.slots[slot]
        if st.phase != "idle":
            raise _lib.CofiError("FrameLoader.begin: slot %d is still %s (release() it after its forward has been collected)" % (slot, st.phase))
        st.h = self.preps[slot].begin(data, img, K, P_Tr, index)
        st.phase = "voxel"

    def _submit(self, st: _Slot):
        h = st.h
        if "event" in h:
            if int(h["cnt_host"][1]):
                raise _lib.CofiError("dataside: the scan spans more than 8192 voxels along an axis")
            nvox = int(h["cnt_host"][0])
        else:
            nvox = h["nvox"]
        h["nvox_final"] = nvox
        st.future = self.pool.submit(draw_frame, h["index"], nvox, self.opt.num_pc, self._amp, self.dataset, NUM_STAGES)
        st.phase = "drawing"

    def poll(self):
        """hand every slot whose voxel count has arrived to the draw workers; never blocks"""
        for st in self.slots:
            if st.phase == "voxel" and ("event" not in st.h or st.h["event"].query()):
                self._submit(st)

    # ------------------------------------------------------------------ stage B
    def _buffers(self, st: _Slot):
        if st.stage is None:
            n, dev = self.opt.num_pc, self.device
            sizes = [n] + [n >> (i + 1) for i in range(NUM_STAGES - 1)]
            st.stage = {"host": [torch.empty(s, dtype=torch.int32, pin_memory=True) for s in sizes] + [torch.empty((4, 4), dtype=torch.float32, pin_memory=True)],
                        "dev": [torch.empty(s, dtype=torch.int32, device=dev) for s in sizes] + [torch.empty((4, 4), dtype=torch.float32, device=dev)],
                        "coarse_host": torch.empty((n >> (NUM_STAGES - 1), 3), dtype=torch.float32, pin_memory=True)}
        return st.stage

    def _stage_b(self, st: _Slot, prep, img_dev, rhw, crop):
        """resample + SE(3) + KNN pyramid + image, all from device-resident inputs: the body of the slot's hipGraph"""
        buf = st.stage["dev"]
        points, feats = prep.resample_transform_dev(st.h["vox"], buf[0], buf[-1])
        pyr = build_pyramid(points, buf[1:-1], int64=False, upsample_k=self.upsample_k)
        pyr["feats"] = feats
        # train mode: the crop is drawn per frame (a kernel argument) and the colour jitter's order / factors with it - the image is then
        # produced by complete() behind the replay, not recorded
        image = prep.image(img_dev, rhw, crop) if self.mode != "train" else None
        st.stage["coarse_host"].copy_(pyr["points"][-1], non_blocking=True)
        return pyr, image

    def complete(self, slot: int) -> Dict:
        """-> the sample of the slot's frame: {'img' (3,H,W), 'pc_data_dict' (points, int32 tables, feats, order: static tensors of the slot),
        'K', 'K_4', 'P', 'index', 'finish_labels'}.  Enqueues on the CURRENT stream (the one begin() was called on)."""
        st, prep, opt, dev = self.slots[slot], self.preps[slot], self.opt, self.device
        if st.phase == "voxel":      # the count has not been polled yet: wait for it (a shallow pipeline ends up here)
            if "event" in st.h:
                st.h["event"].synchronize()
            self._submit(st)
        if st.phase != "drawing":
            raise _lib.CofiError("FrameLoader.complete: slot %d has no frame in preparation" % slot)
        d = st.future.result()
        prep._outstanding = False
        h = st.h
        img, K = h["img"], h["K"]
        buf = self._buffers(st)
        for hb, db, src in zip(buf["host"], buf["dev"], [d["choice"]] + d["sub"] + [d["P"]]):
            hb.copy_(torch.from_numpy(src))
            db.copy_(hb, non_blocking=True)
        s = sampler_from_state(d)   # the generators as the worker's draws left them: crop (train mode), then the label permutations
        K_2, K_4, crop, rhw = dataside.intrinsics_and_crop(K, img.shape[:2], opt, s, self.mode)
        # static image input of the graph: one buffer per image size
        key = (tuple(img.shape), rhw, crop if self.mode != "train" else None, h["vox"].data_ptr())
        ent = st.graphs.get(key)
        if ent is None:
            if len(st.graphs) >= 4:
                st.graphs.clear()   # a new image size / voxel buffer: drop the old captures instead of piling up private pools
            img_static = torch.empty_like(img)
            cap = self._capture_stream or torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream()
            img_static.copy_(img)
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):
                for _ in range(2):   # warm-up: every workspace grown before the capture
                    self._stage_b(st, prep, img_static, rhw, crop)
            cur.wait_stream(cap)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                outs = self._stage_b(st, prep, img_static, rhw, crop)
            ent = st.graphs[key] = (g, outs, img_static)
        g, (pyr, image), img_static = ent
        img_static.copy_(img, non_blocking=True)
        g.replay()
        if self.mode == "train":     # kitti.py:312-314, 329-330: this frame's crop, this frame's jitter
            image = prep.image(img_static, rhw, crop)
            prep.color_jitter(image, *s.color_jitter_params())
        ready = torch.cuda.Event()
        ready.record()
        P = d["P"]
        out = {"img": image, "pc_data_dict": pyr, "index": h["index"], "voxels": h["nvox_final"],
               "K": torch.from_numpy(K_2.astype(np.float32)).to(dev, non_blocking=True), "K_4": torch.from_numpy(K_4.astype(np.float32)).to(dev, non_blocking=True),
               "P": torch.from_numpy(np.linalg.inv(P).astype(np.float32)).to(dev, non_blocking=True)}
        coarse_host = buf["coarse_host"]
        generation = st.generation
        # everything the labels need besides the frame's points: a caller that copies the frame out of the slot (preprocess.FrameStack)
        # releases the slot at once and computes the labels later with labels_from() on its own copy of the points
        out["label_ctx"] = {"P": P, "K_2": K_2, "K_4": K_4, "sampler": s}

        def finish_labels():
            if st.generation != generation:   # the slot's pinned buffer and static tensors belong to another frame by now
                raise _lib.CofiError("FrameLoader: finish_labels() called after the slot was released - call it before release(slot), or "
                                     "use labels_from() on a copy of the frame's points")
            ready.synchronize()
            out.update(self.labels_from(coarse_host.numpy(), pyr["points"][1], pyr["points"][-1], out["label_ctx"]))
            out.pop("finish_labels", None)
            return out

        out["finish_labels"] = finish_labels
        st.phase = "ready"
        return out

    def labels_from(self, coarse_np: np.ndarray, points1: torch.Tensor, points4: torch.Tensor, ctx: Dict) -> Dict:
        """kitti.py:333-420 for one frame from its coarsest-stage points (host array), the device tensors of stages 1 and 4 and the
        `label_ctx` of its sample: the label arrays (device tensors) incl. fine_pc_inline_index = point2node (kitti.py:374)."""
        dev = self.device
        lab = dataside.project_labels(coarse_np, ctx["P"], ctx["K_2"], ctx["K_4"], self.opt, ctx["sampler"], dataset=self.dataset)
        kpt = torch.from_numpy(lab["pc_kpt_idx"]).to(dev)
        res = {"fine_pc_inline_index": ops.nearest_node(points1, points4[kpt].contiguous()).to(torch.int64)}
        for k, v in lab.items():
            res[k] = torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v
        return res

    def release(self, slot: int):
        """the forward that read the slot's tensors has been collected: the slot may begin() another frame"""
        st = self.slots[slot]
        st.phase, st.h, st.future = "idle", None, None
        st.generation += 1
