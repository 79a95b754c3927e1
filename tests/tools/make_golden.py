"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (development container only).

    python tests/tools/make_golden.py [--only micro|tiny|kitti|knn]

Imports /root/reference through tests/tools/ref_shims.py, fills it with the name-keyed
synthetic weights of cofii2p_amd.spec.synth_state_dict and records the reference's outputs on
seeded inputs.  Inputs that are cheap to regenerate deterministically (synthetic frames, KNN
pyramids from the tie-defined C oracle) are stored as seeds + SHA-256; everything else is
stored verbatim.  The fixtures are data: expected outputs of the reference, nothing else.
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import ref_shims  # noqa: E402

from cofii2p_amd.spec import synth_state_dict  # noqa: E402
from cofii2p_amd.synth import make_frame  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def frame_inputs(frame_id, num_points, pyr_seed):
    """The deterministic input chain shared with tests/ (see tests/common.py)."""
    import cofi_oracle as O
    import knn_c

    fr = make_frame(frame_id, num_points=num_points)
    pyr = O.build_pyramid(np.ascontiguousarray(fr.points.T), 5, np.random.RandomState(pyr_seed), knn=knn_c.knn_torch_compatible)
    data = dict(pyr)
    data["feats"] = torch.from_numpy(fr.feats)
    return fr, data


def build_reference_model(norm="gn"):
    net = ref_shims.import_reference()
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(norm=norm).items()}
    opt = ref_shims.reference_options()
    opt.norm = norm   # get_norm(), model/kpconv/modules.py:51-60
    model = net.CoFiI2P(opt)
    model.load_state_dict(sd, strict=True)   # strict: the generated spec has the reference's keys and shapes for this norm
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.eval()
    return net, model, sd


def gen_micro(net, model):
    import importlib

    kpconv_mod = importlib.import_module("model.kpconv.kpconv")
    modules = importlib.import_module("model.kpconv.modules")
    functional = importlib.import_module("model.kpconv.functional")
    tr = importlib.import_module("model.transformer.transformer")
    la = importlib.import_module("model.transformer.linear_attention")
    pe = importlib.import_module("model.transformer.position_encoding")
    g = np.random.default_rng(20240901)
    out = {}

    def rnd(*shape, scale=1.0):
        return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))

    with torch.no_grad():
        # ---- KPConv operator (kpconv.py:79-122) incl. shadow indices and zero-feature rows
        N, M, H, cin, cout = 96, 64, 16, 8, 8
        s_pts = rnd(N, 3, scale=0.3)
        q_pts = s_pts[:M] + rnd(M, 3, scale=0.02)
        idx = torch.from_numpy(g.integers(0, N, (M, H)))
        idx[::7, -3:] = N  # shadow neighbours
        idx[5, :] = N  # a row with no valid neighbour at all
        feats = rnd(N, cin)
        feats[::5] = 0.0  # rows whose feature sum is not > 0
        feats[3] = -feats[3].abs()
        conv = kpconv_mod.KPConv(cin, cout, 15, 0.425, 0.2, bias=True)
        conv.weights.copy_(rnd(15, cin, cout, scale=0.2))
        conv.bias.copy_(rnd(cout, scale=0.1))
        out.update(kp_s_pts=s_pts, kp_q_pts=q_pts, kp_idx=idx, kp_feats=feats, kp_weights=conv.weights.detach().clone(),
                   kp_bias=conv.bias.detach().clone(), kp_kernel_points=conv.kernel_points.clone(),
                   kp_out=conv(feats, q_pts, s_pts, idx))
        # ---- maxpool / nearest_upsample (functional.py:5-21,53-66)
        x = rnd(N, 24)
        out.update(pool_x=x, pool_out=functional.maxpool(x, idx), up_out=functional.nearest_upsample(x, idx))
        # ---- GroupNorm stack mode + UnaryBlock (modules.py:32-49,63-94)
        gn = modules.GroupNorm(32, 64)
        gn.norm.weight.copy_(1 + 0.1 * rnd(64))
        gn.norm.bias.copy_(0.1 * rnd(64))
        x = rnd(77, 64) * 2 + 0.5
        out.update(gn_x=x, gn_w=gn.norm.weight.detach().clone(), gn_b=gn.norm.bias.detach().clone(), gn_out=gn(x))
        un = modules.UnaryBlock(24, 64, "gn", 32)
        un.mlp.weight.copy_(rnd(64, 24, scale=0.2)); un.mlp.bias.copy_(rnd(64, scale=0.1))
        un.norm.norm.weight.copy_(1 + 0.1 * rnd(64)); un.norm.norm.bias.copy_(0.1 * rnd(64))
        x = rnd(50, 24)
        out.update(un_x=x, un_w=un.mlp.weight.detach().clone(), un_b=un.mlp.bias.detach().clone(),
                   un_gw=un.norm.norm.weight.detach().clone(), un_gb=un.norm.norm.bias.detach().clone(), un_out=un(x))
        # ---- FullAttention (linear_attention.py:56-79)
        L, S = 48, 80
        q, k, v = rnd(1, L, 4, 32, scale=0.5), rnd(1, S, 4, 32), rnd(1, S, 4, 32)
        out.update(att_q=q[0], att_k=k[0], att_v=v[0], att_out=la.FullAttention()(q, k, v)[0])
        # ---- LoFTREncoderLayer (transformer.py:43-64) incl. the token-axis Q normalisation
        lay = tr.LoFTREncoderLayer(128, 4)
        for n_, p_ in lay.named_parameters():
            p_.copy_(rnd(*p_.shape, scale=0.08) + (1.0 if n_.endswith("norm1.weight") or n_.endswith("norm2.weight") else 0.0))
        x, src = rnd(1, L, 128), rnd(1, S, 128)
        out.update(lay_x=x[0], lay_src=src[0], lay_out=lay(x, src)[0])
        for n_, p_ in lay.named_parameters():
            out["lay_w_" + n_] = p_.detach().clone()
        # ---- PositionEmbeddingCoordsSine (position_encoding.py:7-50)
        gy, gx = torch.meshgrid(torch.arange(20), torch.arange(64), indexing="ij")
        grid = torch.stack([gy, gx], -1).reshape(1, -1, 2)
        xyz = torch.from_numpy(g.uniform(-80, 80, (1, 200, 3)).astype(np.float32))
        out.update(pe_grid=grid[0], pe_grid_out=pe.PositionEmbeddingCoordsSine(2, 128)(grid)[0], pe_xyz=xyz[0],
                   pe_xyz_out=pe.PositionEmbeddingCoordsSine(3, 128)(xyz)[0])
        # ---- coarse matching (network.py:167-187) with border cases and an exact tie
        Np = 40
        pc = torch.nn.functional.normalize(rnd(128, Np), dim=0)
        im = torch.nn.functional.normalize(rnd(1, 128, 20, 64), dim=1)
        # force chosen pixels: copy point descriptors into specific pixels (incl. border / outside border)
        forced = {0: (0, 0), 1: (2, 2), 2: (18, 62), 3: (19, 63), 4: (10, 1), 5: (1, 30), 6: (18, 63), 7: (9, 33)}
        for pi, (r, c) in forced.items():
            im[0, :, r, c] = pc[:, pi]
        im[0, :, 12, 40] = pc[:, 8]; im[0, :, 12, 41] = pc[:, 8]  # exact tie -> first index
        score = torch.from_numpy(g.uniform(0.5, 1.0, (1, 1, Np)).astype(np.float32))
        score[0, 0, :9] = 0.95
        score[0, 0, 9] = 0.9  # stored as float32(0.9) == threshold in fp32
        xy, sel = net.fine_process(score, pc, im, thrs=0.9)
        out.update(fp_score=score, fp_pc=pc, fp_img=im, fp_xy=xy, fp_sel=sel)
        # ---- point2node / extract_patch (network.py:206-264)
        nodes = rnd(300, 3, scale=10)
        pts = torch.cat([nodes[g.integers(0, 300, 20)] + rnd(20, 3, scale=0.01), rnd(10, 3, scale=10)])
        out.update(p2n_nodes=nodes, p2n_pts=pts, p2n_out=net.point2node(nodes, pts))
        fmap = rnd(1, 8, 80, 256)
        ctr = torch.from_numpy(np.stack([g.integers(2, 63, 12) * 4, g.integers(2, 19, 12) * 4]).astype(np.float32))
        out.update(ep_fmap=fmap[0], ep_ctr=ctr, ep_out=torch.squeeze(net.extract_patch(fmap, ctr)))
        # ---- fine matching: evaluation/eval_all.py:99-105 is not importable (cv2 / dataset imports at
        # eval_all.py:8-13); its six tensor statements are evaluated here with the same torch calls.
        pf = torch.nn.functional.normalize(rnd(12, 64), dim=1)
        pt = torch.nn.functional.normalize(rnd(12, 64, 16), dim=1)
        pt[3, :, 5] = pf[3]; pt[7, :, 15] = pf[7]; pt[9, :, 0] = pf[9]
        dist = torch.squeeze(torch.cosine_similarity(pt.unsqueeze(-1), pf.unsqueeze(-1).unsqueeze(-2)))
        pred = torch.argmax(dist, dim=1)
        fxy = ctr - 2
        fxy[0] = fxy[0] + pred // 4
        fxy[1] = fxy[1] + pred % 4
        out.update(fm_patches=pt, fm_pc=pf, fm_pred=pred, fm_xy=fxy)
        # ---- sub-modules of the assembled model on small inputs
        x = rnd(1, 128, 70)
        out.update(sh_x=x[0], sh_pc_out=model.pc_score_layer(x)[0], sh_img_out=model.img_score_layer(x.reshape(1, 128, 7, 10))[0])
        x = rnd(33, 2048)
        out.update(mlp_x=x, mlp_out=model.pc_feature_layer(x))
        low, skip = rnd(1, 128, 6, 10), rnd(1, 64, 12, 20)
        out.update(ups_low=low[0], ups_skip=skip[0], ups_out=model.img_upsample_1(low, skip)[0])
        img = torch.from_numpy(g.random((1, 3, 64, 96), dtype=np.float32))
        maps = model.img_encoder(img)
        out.update(rn_img=img[0], **{"rn_out%d" % i: m[0] for i, m in enumerate(maps)})
    np.savez_compressed(os.path.join(GOLD, "micro_ops.npz"), **{k: v.numpy() if torch.is_tensor(v) else v for k, v in out.items()})
    print("micro_ops.npz: %d arrays" % len(out))


def run_frame(model, name, frame_id, num_points, pyr_seed, modes):
    fr, data = frame_inputs(frame_id, num_points, pyr_seed)
    img = torch.from_numpy(fr.img)[None]
    out = {"frame_id": frame_id, "num_points": num_points, "pyr_seed": pyr_seed,
           "sha_points": sha(fr.points), "sha_img": sha(fr.img), "sha_feats": sha(fr.feats)}
    for i in range(5):
        out["sha_neighbors%d" % i] = sha(data["neighbors"][i].numpy())
        if i < 4:
            out["sha_subsampling%d" % i] = sha(data["subsampling"][i].numpy())
            out["sha_upsampling%d" % i] = sha(data["upsampling"][i].numpy())
    taps = {}
    hooks = []
    for bname, mod in model.pc_encoder.named_children():
        hooks.append(mod.register_forward_hook(lambda m, i, o, n=bname: taps.__setitem__(n, o.detach())))
    hooks.append(model.transformer.register_forward_hook(lambda m, i, o: taps.__setitem__("transformer", (o[0].detach(), o[1].detach()))))
    hooks.append(model.img_upsample_2.register_forward_hook(lambda m, i, o: taps.__setitem__("img_upsample_2", o.detach())))
    g = np.random.default_rng(99 + frame_id)
    Kp = 16
    kpt = torch.from_numpy(np.stack([g.integers(2, 254, Kp), g.integers(2, 78, Kp)]).astype(np.float32))
    inl = torch.from_numpy(g.integers(0, num_points // 2, Kp))
    out["val_kpt"], out["val_inl"] = kpt.numpy(), inl.numpy()
    names = ("img_desc", "pc_desc", "img_score", "pc_score", "patches", "fine_pc", "center_xy", "coarse_pts")
    with torch.no_grad():
        for mode in modes:
            res = model(data, img, kpt, None, inl, mode)
            for n_, t in zip(names, res):
                if t is not None:
                    out["%s_%s" % (mode, n_)] = t.numpy()
    for h in hooks:
        h.remove()
    for k, v in taps.items():
        if isinstance(v, tuple):
            out["tap_transformer_img"] = v[0][0, ::16].numpy()
            out["tap_transformer_pc"] = v[1][0, ::16].numpy()
        elif v.dim() == 2:
            out["tap_" + k] = v[:: max(1, v.shape[0] // 8)][:8].numpy()  # 8 evenly spaced rows
        else:
            out["tap_" + k] = v[0, :, ::8, ::16].numpy()
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("sha")})


def gen_knn():
    """The reference's OWN neighbour search: model/kpconv/preprocess_data.py `square_distance` (:109-128) + `knn` (:131-143), the
    functions `precompute_point_cloud_cuda` (:145-203) builds every table of the pyramid with.  Recorded per (support, query)
    pair: the reference's index rows and the distances it ranked them by (its own square_distance values at those indices).
    The clouds hold duplicates (sub-sampling with replacement, :58) and lattice ties."""
    import importlib

    ref_shims.import_reference()
    pre = importlib.import_module("model.kpconv.preprocess_data")
    fr = make_frame(3, 2048)
    pts = torch.from_numpy(fr.points)
    rs = np.random.RandomState(0)
    sub = pts[torch.from_numpy(rs.choice(2048, 1024))]          # with replacement: duplicates
    tiny = pts[:100]                                            # fewer support points than k is not a reference case: k = 64 here
    out = {"pts": pts.numpy(), "sub": sub.numpy()}
    cases = {"self": (pts, pts[:192], 128), "down": (pts, sub[:192], 128), "up": (sub, pts[:192], 128), "tiny": (tiny, pts[:64], 64)}
    with torch.no_grad():
        for name, (support, query, k) in cases.items():
            idx = pre.knn(support, query, k)                                          # (Q, k) int64, reference order
            d_all = pre.square_distance(query.unsqueeze(0), support.unsqueeze(0))[0]   # (Q, S)
            out["idx_" + name] = idx.numpy()
            out["dist_" + name] = torch.gather(d_all, 1, idx).numpy()
            out["k_" + name] = k
            out["nq_" + name] = query.shape[0]
    np.savez_compressed(os.path.join(GOLD, "knn_ref.npz"), **out)
    print("knn_ref.npz:", {k_: (v.shape if hasattr(v, "shape") else v) for k_, v in out.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.only == "knn":
        return gen_knn()
    net, model, sd = build_reference_model()
    if args.only is None:
        gen_knn()
    if args.only in (None, "micro"):
        gen_micro(net, model)
    if args.only in (None, "tiny"):
        run_frame(model, "frame_tiny.npz", frame_id=1, num_points=2048, pyr_seed=11, modes=("val", "test"))
    if args.only in (None, "kitti"):
        run_frame(model, "frame_kitti.npz", frame_id=0, num_points=20480, pyr_seed=7, modes=("test",))
    if args.only in (None, "norms"):   # the other two get_norm() configurations of the point encoder (eval mode: BatchNorm = running statistics)
        import json

        for norm in ("bn", "ln"):
            _, m2, _ = build_reference_model(norm)
            # the reference's own state_dict layout for this option (key order, shapes, dtypes): tests/test_host_cpu.py pins the module to it
            spec = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m2.state_dict().items()]
            json.dump(spec, open(os.path.join(GOLD, "state_dict_spec_%s.json" % norm), "w"))
            run_frame(m2, "frame_tiny_%s.npz" % norm, frame_id=1, num_points=2048, pyr_seed=11, modes=("test",))


if __name__ == "__main__":
    main()
