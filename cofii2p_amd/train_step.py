"""The caller side of one optimisation step - train.py:186-285 - on tensors that are already on the device.

train.py is a script (argparse, dataset construction and a TensorBoard writer at import time); what it does between the forward and
`loss.backward()` is restated here with the reference's own tensor expressions so that tests and `bench.py --train` drive this module
exactly the way train.py drives the reference's: forward(mode='train') -> key-point gathers -> projection -> correspondence mask ->
desc_loss + overlap_loss + fine_circle_loss (cofii2p_amd.loss: HIP kernels with analytic gradients) -> backward.

`GraphedTrainStep` is the same step recorded ONCE into a hipGraph (forward, losses, backward, Adam: ~3 800 launches) and replayed per
frame: the eager step is bound by the Python thread that issues those launches, the replay by the kernels.  Possible because every
KITTI / nuScenes training frame has the same tensor signature (fixed point counts per pyramid stage, k = 128 tables, num_kpt labels)
and nothing in the step reads a device value on the host.
"""
from typing import Dict, List, Optional, Tuple

import torch

from . import loss as L


def step_losses(model, pc_data_dict: Dict, img: torch.Tensor, batch: Dict[str, torch.Tensor], opt, mode: str = "train") -> Tuple:
    """train.py:224-283.  `batch`: K_4 (3,3), P (4,4), pc_kpt_idx / pc_outline_idx / coarse_img_kpt_idx (num_kpt,) int64,
    fine_center_kpt_coors / fine_xy (2, num_kpt) int64, fine_pc_inline_index (num_kpt,) int64 - the fields data/kitti.py:305-420 puts
    into a training sample.  `opt`: dist_thres, pos_margin, neg_margin.  -> (outputs, mask, (loss_desc, loss_coarse, loss_fine))."""
    dev = img.device
    K = batch["pc_kpt_idx"].numel()
    outs = model(pc_data_dict, img, batch["fine_center_kpt_coors"], batch["fine_xy"], batch["fine_pc_inline_index"], mode)
    img_features, pc_features, coarse_img_score, coarse_pc_score, fine_patch, fine_pc_feat = outs[:6]
    pc_kpt_idx, pc_outline_idx = batch["pc_kpt_idx"], batch["pc_outline_idx"]
    pc_features_inline = torch.gather(pc_features, index=pc_kpt_idx.expand(pc_features.size(0), K), dim=-1)                    # train.py:234
    pc_xyz_inline = torch.gather(pc_data_dict["points"][-1].T, index=pc_kpt_idx.unsqueeze(0).expand(3, K), dim=-1)          # train.py:238
    img_features_flatten = img_features.contiguous().view(img_features.size(1), -1)                                         # train.py:240
    H8, W8 = img_features.shape[2:]
    img_x = torch.linspace(0, W8 - 1, W8, device=dev).view(1, -1).expand(H8, W8).unsqueeze(0)                               # train.py:219-222
    img_y = torch.linspace(0, H8 - 1, H8, device=dev).view(-1, 1).expand(H8, W8).unsqueeze(0)
    img_xy_flatten = torch.cat((img_x, img_y), dim=0).contiguous().view(2, -1)
    cidx = batch["coarse_img_kpt_idx"]
    img_features_flatten_inline = torch.gather(img_features_flatten, index=cidx.unsqueeze(0).expand(img_features_flatten.size(0), K), dim=-1)
    img_xy_flatten_inline = torch.gather(img_xy_flatten, index=cidx.unsqueeze(0).expand(2, K), dim=-1)
    P, K_4 = batch["P"], batch["K_4"]
    proj = torch.mm(K_4, (torch.mm(P[0:3, 0:3], pc_xyz_inline) + P[0:3, 3:]))                                              # train.py:248
    pc_xy = proj[0:2, :] / proj[2:, :]
    mask = (torch.sqrt(torch.sum(torch.square(img_xy_flatten_inline.unsqueeze(-1) - pc_xy.unsqueeze(-2)), dim=0)) <= opt.dist_thres).float()
    loss_desc, _dists = L.desc_loss(dev, img_features_flatten_inline, pc_features_inline, mask, pos_margin=opt.pos_margin, neg_margin=opt.neg_margin)
    s_in = torch.squeeze(coarse_pc_score[:, :, pc_kpt_idx])                                                                 # train.py:256-257
    s_out = torch.squeeze(coarse_pc_score[:, :, pc_outline_idx])
    loss_coarse = L.overlap_loss(dev, s_in, s_out)
    rel = batch["fine_xy"] - batch["fine_center_kpt_coors"] + 2                                                             # train.py:268-269
    rel_index = rel[1, :] * 4 + rel[0, :]
    loss_fine = L.fine_circle_loss(dev, fine_patch, fine_pc_feat, rel_index, K)
    return outs, mask, (loss_desc, loss_coarse, loss_fine)


def train_step(model, optimizer, pc_data_dict, img, batch, opt):
    """train.py:188-286: model.train(); zero_grad; forward; losses; backward; optimizer.step().  -> the three loss values (tensors)."""
    model.train()
    optimizer.zero_grad()
    _outs, _mask, (l_desc, l_coarse, l_fine) = step_losses(model, pc_data_dict, img, batch, opt)
    (l_desc + l_coarse + l_fine).backward()
    optimizer.step()
    return l_desc.detach(), l_coarse.detach(), l_fine.detach()


_PYRAMID_LISTS = ("points", "neighbors", "subsampling", "upsampling")
_BATCH_KEYS = (("K_4", "K_4"), ("P", "P"), ("pc_kpt_idx", "pc_kpt_idx"), ("pc_outline_idx", "pc_outline_idx"), ("coarse_img_kpt_idx", "coarse_img_kpt_idx"),
               ("fine_center_kpt_coors", "fine_center_kpt_coors"), ("fine_xy", "fine_xy_coors"), ("fine_pc_inline_index", "fine_pc_inline_index"))


def batch_from_sample(sample: Dict, device=None) -> Tuple[Dict, torch.Tensor, Dict[str, torch.Tensor]]:
    """train.py:192-217: what one optimisation step reads out of a data-side sample - `FramePreparer.prepare`, `FrameLoader.complete`
    (after `finish_labels()`), or a batch of 1 of the reference's DataLoader - as (pc_data_dict, img (1, 3, H, W), batch) for
    `step_losses` / `train_step` / `GraphedTrainStep`.  Leading batch dimensions of 1 are squeezed the way train.py does it."""
    dev = torch.device(device) if device is not None else sample["img"].device
    sq = lambda t: torch.squeeze(t.to(dev), 0) if t.dim() and t.shape[0] == 1 else t.to(dev)
    pc = sample["pc_data_dict"]
    pc_data_dict = {k: [sq(t) for t in pc[k]] for k in _PYRAMID_LISTS}
    pc_data_dict["feats"] = sq(pc["feats"])
    img = sample["img"].to(dev)
    img = img[None] if img.dim() == 3 else img
    return pc_data_dict, img, {k: sq(sample[src]) for k, src in _BATCH_KEYS}


class GraphedTrainStep:
    """train.py:188-286 (zero_grad; forward(mode='train'); three losses; backward; optimizer.step()) as one hipGraph.

        step = GraphedTrainStep(model, torch.optim.Adam(params, lr=opt.lr, capturable=True), opt)
        for data in loader:
            losses = step(pc_data_dict, img, batch)       # (3,) device tensor: desc, coarse (overlap), fine - values of THIS step

    The first call with a given tensor signature runs the step eagerly (a real optimisation step; it also initialises the optimizer's
    state), the second records the graph, every later one copies the frame into the graph's static inputs (index tables as int32) and
    replays it.  Bit-identical to the eager step with the same optimizer (tests/test_train_gpu.py).

    * The optimizer must be capturable (its step counters live on the device): anything else would bake this step's bias
      correction into the recording.  The learning rate is kept as a device scalar the recording reads; train.py:326-330's
      `param_group['lr'] = current_lr` keeps working (the new value is written into that scalar before the next replay).
    * `validate=True` checks on the host that the 4 x 4 patches stay inside the feature map (the reference's assertion,
      network.py:222) before a frame is launched - one small device-to-host read per step; `validate=False` leaves it out.
    * Do not keep the outputs / loss tensors of an EAGER step of the same module alive across the recording call: their autograd
      graph pins gradient-accumulation nodes to the eager stream, which breaks a capture (torch warns, HIP aborts).
    * After replays the module's packed inference weights are refreshed on the next inference forward, as after eager steps.
    """

    def __init__(self, model, optimizer, opt, validate: bool = True):
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep needs a capturable optimizer (torch.optim.Adam(..., capturable=True)): a host-side "
                                 "step counter would be frozen into the recording")
        self.model, self.optimizer, self.opt, self.validate = model, optimizer, opt, validate
        self._sig = None
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._warm = False
        self._static: Optional[Tuple[Dict, torch.Tensor, Dict[str, torch.Tensor]]] = None
        self._losses: Optional[torch.Tensor] = None
        self._lr: List[torch.Tensor] = []
        self.replays = 0

    def _capture_stream(self, dev) -> torch.cuda.Stream:
        if getattr(self, "_cap", None) is None or self._cap.device != torch.device(dev):
            self._cap = torch.cuda.Stream(device=dev)
        return self._cap

    # ---- learning rate: a device scalar per parameter group
    def _sync_lr(self, dev):
        if not self._lr:
            self._lr = [torch.full((), float(g["lr"]), dtype=torch.float32, device=dev) for g in self.optimizer.param_groups]
        for g, t in zip(self.optimizer.param_groups, self._lr):
            if g["lr"] is not t:
                t.fill_(float(g["lr"]))
                g["lr"] = t

    # ---- static inputs
    @staticmethod
    def _signature(pc, img, batch):
        sig = [tuple(img.shape)]
        for k in _PYRAMID_LISTS:
            sig += [tuple(t.shape) for t in pc[k]]
        sig.append(tuple(pc["feats"].shape))
        sig += [(k, tuple(batch[k].shape), str(batch[k].dtype)) for k in sorted(batch)]
        return tuple(sig)

    def _allocate(self, pc, img, batch, dev):
        spc = {"points": [torch.empty(t.shape, dtype=torch.float32, device=dev) for t in pc["points"]],
               "feats": torch.empty(pc["feats"].shape, dtype=torch.float32, device=dev)}
        for k in _PYRAMID_LISTS[1:]:
            spc[k] = [torch.empty(t.shape, dtype=torch.int32, device=dev) for t in pc[k]]
        simg = torch.empty(img.shape, dtype=torch.float32, device=dev)
        sbatch = {k: torch.empty(v.shape, dtype=v.dtype, device=dev) for k, v in batch.items()}
        self._static = (spc, simg, sbatch)

    def _stage(self, pc, img, batch):
        spc, simg, sbatch = self._static
        for k in _PYRAMID_LISTS:
            for dst, src in zip(spc[k], pc[k]):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)          # int64 tables narrow to int32 here (indices < 2^31: at most N rows)
        if spc["feats"].data_ptr() != pc["feats"].data_ptr():
            spc["feats"].copy_(pc["feats"])
        if simg.data_ptr() != img.data_ptr():
            simg.copy_(img)
        for k, dst in sbatch.items():
            if dst.data_ptr() != batch[k].data_ptr():
                dst.copy_(batch[k])

    def static_inputs(self):
        """(pc_data_dict, img, batch) the recording reads: a loader may write the next frame straight into them (then the copies of
        `__call__` are skipped for the tensors it is handed back)."""
        return self._static

    def _check_patches(self, img):
        ctr = self._static[2]["fine_center_kpt_coors"]
        H2, W2 = (img.shape[-2] + 1) // 2, (img.shape[-1] + 1) // 2     # the stride-2 stem (imagenet.py:21)
        lt = torch.floor(ctr.to(torch.float32) - 2.0)
        bad = (lt < 0).any() | (lt[0] + 3 >= W2).any() | (lt[1] + 3 >= H2).any()
        if bool(bad):
            raise AssertionError("patch leaves the feature map (network.py:222)")

    def _record(self):
        import gc

        gc.collect()    # autograd graphs of earlier eager steps that only a reference cycle keeps alive
        spc, simg, sbatch = self._static
        self.optimizer.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        # captured on the stream the warm-up step ran on: every per-(stream, slot) scratch buffer of cofii2p_amd.ops (and the side stream of
        # the forward's branches) was grown there, in the ordinary allocator pool - nothing is first allocated inside the graph's private pool
        with torch.cuda.graph(graph, stream=self._capture_stream(spc["feats"].device)):
            _o, _m, (l_desc, l_coarse, l_fine) = step_losses(self.model, spc, simg, sbatch, self.opt)
            (l_desc + l_coarse + l_fine).backward()
            self.optimizer.step()
            self._losses = torch.stack([l_desc.detach(), l_coarse.detach(), l_fine.detach()])
        self._graph = graph

    def __call__(self, pc_data_dict, img, batch) -> torch.Tensor:
        dev = img.device
        if dev.type != "cuda":
            raise ValueError("GraphedTrainStep: the frame must be on the GPU")
        sig = self._signature(pc_data_dict, img, batch)
        if sig != self._sig:        # another frame shape: new static inputs, new recording (the old one is dropped)
            self._sig, self._graph, self._warm = sig, None, False
            self._allocate(pc_data_dict, img, batch, dev)
        self._sync_lr(dev)
        self._stage(pc_data_dict, img, batch)
        if self.validate:
            self._check_patches(img)
        self.model.train()
        if self._graph is None and not self._warm:
            self._warm = True
            spc, simg, sbatch = self._static
            cap, cur = self._capture_stream(dev), torch.cuda.current_stream(dev)
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):   # the eager warm-up step runs where the recording will be captured (see _record)
                self.optimizer.zero_grad()
                _o, _m, (l_desc, l_coarse, l_fine) = step_losses(self.model, spc, simg, sbatch, self.opt)
                (l_desc + l_coarse + l_fine).backward()
                self.optimizer.step()
                out = torch.stack([l_desc.detach(), l_coarse.detach(), l_fine.detach()])
            cur.wait_stream(cap)
            out.record_stream(cur)
            return out
        if self._graph is None:
            self._record()
        self._graph.replay()
        self.replays += 1
        for m in self.model.modules():      # parameters changed behind their version counters: CoFiI2P._pack() looks at this count
            if hasattr(m, "_replayed_steps"):
                m._replayed_steps += 1
        return self._losses.clone()
