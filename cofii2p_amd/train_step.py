"""The caller side of one optimisation step - train.py:186-285 - on tensors that are already on the device.

train.py is a script (argparse, dataset construction and a TensorBoard writer at import time); what it does between the forward and
`loss.backward()` is restated here with the reference's own tensor expressions so that tests and `bench.py --train` drive this module
exactly the way train.py drives the reference's: forward(mode='train') -> key-point gathers -> projection -> correspondence mask ->
desc_loss + overlap_loss + fine_circle_loss (cofii2p_amd.loss: HIP kernels with analytic gradients) -> backward.
"""
from typing import Dict, Tuple

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
