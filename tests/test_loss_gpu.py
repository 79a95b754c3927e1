"""Row f3, first part: the HIP loss kernels (csrc/loss.hip through cofii2p_amd/loss.py's autograd Functions) against the REFERENCE's own
model/loss.py - loss values, the returned `dists`, and the gradients torch.autograd derives from the reference's expressions
(tests/golden/loss_ref.npz, recorded by tests/tools/make_golden_loss.py).  Needs a real MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from common import load_golden  # noqa: E402

DEV = "cuda:0"
TOL = 2e-5   # fp32 sums in another order than torch's


@pytest.fixture(scope="module")
def lg():
    return load_golden("loss_ref.npz")


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_() if grad else t


def close(a, b, tol=TOL):
    np.testing.assert_allclose(a.detach().cpu().numpy(), np.asarray(b), rtol=tol, atol=tol)


@pytest.mark.parametrize("tag", ["kitti", "nuscenes", "odd"])
def test_desc_loss_value_dists_and_gradients(lg, tag):
    from cofii2p_amd.loss import desc_loss

    img, pc = G(lg[tag + "_img"], True), G(lg[tag + "_pc"], True)
    loss, dists = desc_loss(DEV, img, pc, G(lg[tag + "_mask"]), pos_margin=0.2, neg_margin=1.8)
    close(loss, lg[tag + "_desc_loss"])
    close(dists, lg[tag + "_dists"], 1e-5)
    (3.0 * loss).backward()          # a non-unit upstream gradient
    close(img.grad / 3.0, lg[tag + "_desc_gimg"])
    close(pc.grad / 3.0, lg[tag + "_desc_gpc"])


@pytest.mark.parametrize("tag", ["kitti", "nuscenes", "odd"])
def test_overlap_loss_value_and_gradients(lg, tag):
    from cofii2p_amd.loss import overlap_loss

    s_in, s_out = G(lg[tag + "_sin"], True), G(lg[tag + "_sout"], True)
    loss = overlap_loss(DEV, s_in, s_out)
    close(loss, lg[tag + "_overlap_loss"])
    loss.backward()
    close(s_in.grad, lg[tag + "_overlap_gin"])
    close(s_out.grad, lg[tag + "_overlap_gout"])


@pytest.mark.parametrize("tag", ["kitti", "nuscenes", "odd"])
def test_fine_circle_loss_value_and_gradients(lg, tag):
    from cofii2p_amd.loss import fine_circle_loss

    patches, fpc = G(lg[tag + "_patches"], True), G(lg[tag + "_fpc"], True)
    K = patches.shape[0]
    loss = fine_circle_loss(DEV, patches, fpc, G(lg[tag + "_rel"]), num_kpt=K)
    close(loss, lg[tag + "_fine_loss"])
    loss.backward()
    close(patches.grad, lg[tag + "_fine_gpatches"])
    close(fpc.grad, lg[tag + "_fine_gpc"])


def test_losses_on_the_forward_outputs_train_py_shape():
    """train.py:233-282 with the reference's own gather statements around the HIP forward (under no_grad: the network's backward is not
    built) and the HIP losses: finite values, gradients arrive at the forward's outputs, the sum is what train.py:283 adds up."""
    import bench
    from cofii2p_amd.loss import desc_loss, fine_circle_loss, overlap_loss
    from cofii2p_amd.network import CoFiI2P

    model = CoFiI2P(bench.Opt()).to(DEV)
    pyr, img, fr = bench.make_inputs(torch.device(DEV), [0], 20480)[0]
    K = 64
    g = torch.Generator().manual_seed(1)
    pc_kpt_idx = torch.randperm(1280, generator=g)[:K].to(DEV)
    pc_outline_idx = torch.randperm(1280, generator=g)[:K].to(DEV)
    coarse_img_kpt_idx = torch.randint(0, 1280, (K,), generator=g).to(DEV)
    kpt = torch.stack([torch.randint(2, 254, (K,), generator=g), torch.randint(2, 78, (K,), generator=g)]).float().to(DEV)
    with torch.no_grad():
        img_features, pc_features, _, coarse_pc_score, patch, fine_pc, _, _ = model(pyr, img, kpt, None, pc_kpt_idx, "val")
    img_features, pc_features, coarse_pc_score = img_features.requires_grad_(), pc_features.requires_grad_(), coarse_pc_score.requires_grad_()
    patch, fine_pc = patch.requires_grad_(), fine_pc.requires_grad_()
    pc_in = torch.gather(pc_features, index=pc_kpt_idx.expand(pc_features.size(0), K), dim=-1)                      # train.py:236
    img_flat = img_features.contiguous().view(img_features.size(1), -1)                                             # train.py:242
    img_in = torch.gather(img_flat, index=coarse_img_kpt_idx.unsqueeze(0).expand(img_flat.size(0), K), dim=-1)      # train.py:246
    mask = torch.eye(K, device=DEV)
    loss_desc, dists = desc_loss(DEV, img_in, pc_in, mask, pos_margin=0.2, neg_margin=1.8)
    loss_coarse = overlap_loss(DEV, torch.squeeze(coarse_pc_score[:, :, pc_kpt_idx]), torch.squeeze(coarse_pc_score[:, :, pc_outline_idx]))
    loss_fine = fine_circle_loss(DEV, patch, fine_pc, torch.randint(0, 16, (K,), generator=g).to(DEV), K)
    loss = loss_desc + loss_coarse + loss_fine                                                                      # train.py:283
    loss.backward()
    assert torch.isfinite(loss) and dists.shape == (K, K)
    for t in (img_features, pc_features, coarse_pc_score, patch, fine_pc):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0
