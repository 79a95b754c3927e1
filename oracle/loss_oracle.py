"""CPU restatement of the reference's training losses (model/loss.py) - TEST INFRASTRUCTURE ONLY: imported by tests/ and
`__graft_entry__.smoke()`, never by the product (cofii2p_amd/loss.py runs HIP kernels with analytic gradients).

Plain differentiable torch expressions, each citing the lines it follows; pinned against fixtures recorded from the reference's own
functions and its autograd (tests/golden/loss_ref.npz, tests/test_oracle_golden.py)."""
import torch
import torch.nn.functional as F


def desc_loss(img_features, pc_features, mask, pos_margin=0.1, neg_margin=1.4, log_scale=10.0):
    """loss.py:69-93.  img / pc features (C, K) with unit columns, mask (K, K) -> (mean loss, dists)."""
    neg_mask = 1.0 - mask
    dists = 1.0 - torch.sum(img_features.unsqueeze(-1) * pc_features.unsqueeze(-2), dim=0)       # :73
    pos = dists - 1e5 * neg_mask                                                                   # :75
    pos_weight = (pos - pos_margin).detach().clamp_min(0.0)                                        # :76-77: a constant of the graph
    lse_pos_row = torch.logsumexp(log_scale * (pos - pos_margin) * pos_weight, dim=-1)             # :79
    lse_pos_col = torch.logsumexp(log_scale * (pos - pos_margin) * pos_weight, dim=-2)
    neg = dists + 1e5 * mask                                                                       # :82
    neg_weight = (neg_margin - neg).detach().clamp_min(0.0)
    lse_neg_row = torch.logsumexp(log_scale * (neg_margin - neg) * neg_weight, dim=-1)
    lse_neg_col = torch.logsumexp(log_scale * (neg_margin - neg) * neg_weight, dim=-2)
    loss = F.softplus(lse_pos_row + lse_neg_row) / log_scale + F.softplus(lse_pos_col + lse_neg_col) / log_scale   # :89-91
    return loss.mean(), dists


def overlap_loss(inline_score, outline_score):
    """loss.py:53-60: nn.BCELoss() (mean; log clamped at -100 as torch does) of [inline | outline] against [1 | 0]."""
    score = torch.cat([inline_score.reshape(-1), outline_score.reshape(-1)])
    label = torch.cat([torch.ones(inline_score.numel()), torch.zeros(outline_score.numel())]).to(score)
    return -(label * torch.log(score).clamp_min(-100.0) + (1.0 - label) * torch.log(1.0 - score).clamp_min(-100.0)).mean()


def fine_circle_loss(fine_img_feature, fine_pc_feature, relative_index, m=0.2, gamma=5.0):
    """loss.py:9-51.  patches (K, C, 4, 4), point descriptors (K, C), index of the true pixel (K,) in 0..15."""
    K = fine_img_feature.shape[0]
    flat = fine_img_feature.reshape(K, fine_img_feature.shape[1], 16)
    dist = torch.cosine_similarity(flat, fine_pc_feature.unsqueeze(-1), dim=1)                     # :17 (eps 1e-8), (K, 16)
    pos = torch.zeros((K, 16), dtype=dist.dtype)
    pos[torch.arange(K), relative_index.long()] = 1.0                                             # :19-22
    neg = 1.0 - pos
    sp, sn = dist * pos, dist * neg
    ap = torch.relu(-sp.detach() + pos + pos * m)                                                  # :38
    an = torch.relu(sn.detach() + neg * m)                                                         # :40
    logit_p = -ap * (sp - pos * (1.0 - m)) * gamma                                                 # :44
    logit_n = an * (sn - neg * m) * gamma
    loss_p = torch.sum(torch.exp(logit_p) * pos, dim=-1)
    loss_n = torch.sum(torch.exp(logit_n) * neg, dim=-1)
    return torch.mean(torch.log(1.0 + loss_n * loss_p))                                            # :50
