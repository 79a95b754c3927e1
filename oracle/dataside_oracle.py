"""CPU restatement (numpy) of the DATA SIDE of one KITTI frame: everything kitti_pc_img_dataset.__getitem__ does between the
disk read and the model call (/root/reference/data/kitti.py:259-393).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product (cofii2p_amd/dataside.py runs the HIP kernels).

Pinned against the reference: tests/golden/dataside_ref.npz holds the outputs of the reference's own __getitem__ run in the
development container (tests/tools/make_golden_dataside.py) for everything that is numpy / torch in the reference: the calibration
transform, downsample_np, generate_random_transform, the SE(3) application, the random half sub-sampling, the intrinsics scaling /
cropping, the coarse / fine label projection and point2node.
PARITY UNPINNED for the two third-party calls the image lacks: open3d `voxel_down_sample` (restated below from open3d 0.17
geometry/PointCloud.cpp VoxelDownSample; output order differs: ascending voxel index here, unordered_map order there) and
`cv2.resize(INTER_LINEAR)` on uint8 (restated from OpenCV 4.x imgproc/resize.cpp, generic fixed-point path: 11-bit coefficients,
FixedPtCast<int, uchar, 22>).  In the golden run those two calls are served by the functions below.
"""
import random

import numpy as np

NUM_STAGES = 5


def frame_seed(index: int) -> int:
    """kitti.py:261-264."""
    (seed,) = np.random.SeedSequence([index]).generate_state(1)
    return int(seed)


def calib_matrices(lines: dict):
    """kitti.py:24-63 (KittiCalibHelper.read_calib_files) for one sequence: {'Tr': 4x4, 'P2': 4x4, 'P2_K': 3x3, ...} float32."""
    out = {}
    for key, text in lines.items():
        mat = np.array([float(v) for v in text.split()], dtype=np.float64).reshape(3, 4).astype(np.float32)
        if key == "Tr":
            P = np.identity(4, dtype=np.float32)
            P[0:3, :] = mat
            out[key] = P
        else:
            K = mat[0:3, 0:3]
            out[key + "_K"] = K
            fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
            tz = mat[2, 3]
            tx = (mat[0, 3] - cx * tz) / fx
            ty = (mat[1, 3] - cy * tz) / fy
            P = np.identity(4, dtype=np.float32)
            P[0:3, 3] = np.asarray([tx, ty, tz], dtype=np.float32)
            out[key] = P
    return out


def rigid(P, x, translate=True):
    """np.dot(P[0:3,0:3], x) + P[0:3,3:] (kitti.py:276-277, 286-287) on float32 (3, N), restated as separate float32 multiplies and
    adds in the order k = 0, 1, 2 (what the HIP kernels compute; BLAS may fuse or reorder: compared with the reference to 1e-5)."""
    P = P.astype(np.float32)
    x = x.astype(np.float32)
    out = np.empty_like(x)
    for a in range(3):
        s = np.float32(0) + P[a, 0] * x[0]
        s = s + P[a, 1] * x[1]
        s = s + P[a, 2] * x[2]
        out[a] = s + P[a, 3] if translate else s
    return out


def voxel_down_sample(pc, intensity, sn, voxel=0.1):
    """kitti.py:145-166 with open3d's VoxelDownSample: voxel index = floor((p - (min_bound - voxel / 2)) / voxel) in double,
    double means of points / colours / normals per voxel, colours = [intensity / max, 0, 0].  (3,N), (1,N), (3,N) float32 ->
    (3,M), (1,M), (3,M) float32 in ascending (ix, iy, iz) order."""
    pts = pc.T.astype(np.float64)
    imax = np.max(intensity)
    col = (intensity[0] / imax).astype(np.float64)   # float32 division, widened when stored into the colour array
    nrm = sn.T.astype(np.float64)
    minb = pts.min(0) - voxel * 0.5
    vidx = np.floor((pts - minb) / voxel).astype(np.int64)
    key = (vidx[:, 0] << 26) | (vidx[:, 1] << 13) | vidx[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    cnt = np.diff(np.r_[heads, len(ks)]).astype(np.float64)

    def seg_mean(v):   # sequential double sums in original point order inside each voxel, then * (1 / count)
        v = v[order]
        out = np.zeros(len(heads))
        pos = heads.copy()
        live = np.arange(len(heads))
        end = np.r_[heads[1:], len(ks)]
        while len(live):
            out[live] += v[pos[live]]
            pos[live] += 1
            live = live[pos[live] < end[live]]
        return out * (1.0 / cnt)

    P = np.stack([seg_mean(pts[:, a]) for a in range(3)], 0).astype(np.float32)
    I = (seg_mean(col).astype(np.float32) * imax)[None, :]
    S = np.stack([seg_mean(nrm[:, a]) for a in range(3)], 0).astype(np.float32)
    return P, I.astype(np.float32), S


def downsample_choice(n, num_pc, rs):
    """kitti.py:168-176: indices that bring n points to exactly num_pc (rs = the global numpy RandomState of the reference)."""
    if n >= num_pc:
        return rs.choice(n, num_pc, replace=False)
    fix = np.asarray(range(n))
    while n + fix.shape[0] < num_pc:
        fix = np.concatenate((fix, np.asarray(range(n))), axis=0)
    return np.concatenate((fix, rs.choice(n, num_pc - fix.shape[0], replace=False)), axis=0)


def angles2rotation_matrix(angles):
    """kitti.py:203-214: R = Rz Ry Rx."""
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(angles[0]), -s(angles[0])], [0, s(angles[0]), c(angles[0])]])
    Ry = np.array([[c(angles[1]), 0, s(angles[1])], [0, 1, 0], [-s(angles[1]), 0, c(angles[1])]])
    Rz = np.array([[c(angles[2]), -s(angles[2]), 0], [s(angles[2]), c(angles[2]), 0], [0, 0, 1]])
    return np.dot(Rz, np.dot(Ry, Rx))


def random_transform(rnd, amp):
    """kitti.py:216-235; rnd = random.Random (the reference's global `random`), amp = (tx, ty, tz, Rx, Ry, Rz) amplitudes
    (data/options.py:33-38)."""
    t = [rnd.uniform(-amp[0], amp[0]), rnd.uniform(-amp[1], amp[1]), rnd.uniform(-amp[2], amp[2])]
    angles = [rnd.uniform(-amp[3], amp[3]), rnd.uniform(-amp[4], amp[4]), rnd.uniform(-amp[5], amp[5])]
    P = np.identity(4, dtype=np.float32)
    P[0:3, 0:3] = angles2rotation_matrix(angles)
    P[0:3, 3] = t
    return P


def camera_matrix_scaling(K, s):
    """kitti.py:188-191."""
    Ks = s * K
    Ks[2, 2] = 1
    return Ks


def camera_matrix_cropping(K, dx, dy):
    """kitti.py:182-186."""
    Kc = np.copy(K)
    Kc[0, 2] -= dx
    Kc[1, 2] -= dy
    return Kc


def resize_linear_u8(img, dw, dh):
    """cv2.resize(img, (dw, dh), interpolation=INTER_LINEAR) for uint8 HWC (kitti.py:306-309) as OpenCV's fixed-point path computes it
    (imgproc/resize.cpp, published algorithm; cv2 itself is absent from this image -> parity with the library UNPINNED, see
    tests/test_thirdparty_gpu.py): 11-bit coefficients (cvRound), horizontal pass exact in int32, vertical pass as the uchar
    specialisation of VResizeLinear: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 - each term truncated."""
    sh, sw = img.shape[:2]

    def coef(dsize, ssize):
        scale = np.float32(float(ssize) / dsize)
        d = np.arange(dsize)
        f = ((d + 0.5) * np.float64(scale) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= ssize - 1
        f[hi], s[hi] = 0, ssize - 1
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, ssize - 1), a0, a1

    x0, x1, ax0, ax1 = coef(dw, sw)
    y0, y1, ay0, ay1 = coef(dh, sh)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]
    v = (((ay0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((ay1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# train-mode image augmentation: torchvision ColorJitter on a PIL image (kitti.py:193-201, nuscenes.py:109-117)
# --------------------------------------------------------------------------------------
# torchvision (absent from this image; the reference pins no version) applies, in a random order, adjust_brightness / _contrast /
# _saturation / _hue of transforms/_functional_pil.py: PIL.ImageEnhance.Brightness / Contrast / Color = Image.blend(degenerate, img,
# factor), and a uint8 shift of the H plane of img.convert("HSV").  PIL itself IS in the image: tests/test_dataside_cpu.py holds the
# restatement below to PIL's own results bit for bit.  (torchvision draws order and factors from the unseeded torch generator; the
# device-side loader draws them from the frame seed instead - jitter_params - so that a frame is reproducible.)


def jitter_params(seed: int):
    """order of the four operations (0 brightness, 1 contrast, 2 saturation, 3 hue) and their factors: ColorJitter((0.8, 1.2), (0.8, 1.2),
    (0.8, 1.2), (-0.1, 0.1)) (kitti.py:194-199), drawn from the frame seed."""
    rs = np.random.RandomState((int(seed) + 0x9E3779B9) % (1 << 32))
    order = [int(v) for v in rs.permutation(4)]
    fb, fc, fs = (float(v) for v in rs.uniform(0.8, 1.2, 3))
    fh = float(rs.uniform(-0.1, 0.1))
    return order, fb, fc, fs, fh


def _gray_u8(img):
    """PIL convert("L") (ITU-R 601-2 luma, Convert.c L24): (19595 R + 38470 G + 7471 B + 0x8000) >> 16"""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def _blend_u8(deg, img, alpha):
    """PIL Image.blend(degenerate, img, alpha) (Blend.c): float32 deg + alpha (img - deg), truncated; clipped first when alpha is outside [0, 1]"""
    t = deg.astype(np.float32) + np.float32(alpha) * (img.astype(np.float32) - deg.astype(np.float32))
    if not 0.0 <= alpha <= 1.0:
        t = np.clip(t, 0, 255)
    return t.astype(np.int64).astype(np.uint8)


def _rgb2hsv_u8(img):
    """PIL convert("HSV") (Convert.c rgb2hsv_row): float ratios, the hue sum and h / 6 + 1 in double (C literals), (int)(x * 255.0)"""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = cr / maxc.astype(np.float32)
        rc, gc, bc = ((maxc - c).astype(np.float32) / cr for c in (r, g, b))
        rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)
        h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
        h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    flat = minc == maxc
    return np.where(flat, 0, uh).astype(np.uint8), np.where(flat, 0, us).astype(np.uint8), maxc.astype(np.uint8)


def _hsv2rgb_u8(h, s, v):
    """PIL HSV -> RGB (Convert.c hsv2rgb): i = floor(h 6 / 255), f the remainder (float), p / q / t = round(v (1 - s' ...)) in double"""
    hf = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int64)
    f = (hf - i.astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vf = v.astype(np.float64)
    rnd = lambda x: np.clip(np.floor(x + 0.5).astype(np.int64), 0, 255)   # C round() of a non-negative value
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    vv, sel = v.astype(np.int64), i % 6
    r, g, b = np.choose(sel, [vv, q, p, p, t, vv]), np.choose(sel, [t, vv, vv, q, p, p]), np.choose(sel, [p, p, t, vv, vv, q])
    z = s == 0
    return np.stack([np.where(z, vv, r), np.where(z, vv, g), np.where(z, vv, b)], -1).astype(np.uint8)


def color_jitter(img, order, fb, fc, fs, fh):
    """(H, W, 3) uint8 -> (H, W, 3) uint8: the four ColorJitter operations in `order` (0 brightness, 1 contrast, 2 saturation, 3 hue)."""
    for op in order:
        if op == 0:
            img = _blend_u8(np.zeros_like(img), img, fb)
        elif op == 1:
            gm = _gray_u8(img)
            mean = int(gm.astype(np.float64).sum() / gm.size + 0.5)        # int(ImageStat.Stat(gray).mean[0] + 0.5)
            img = _blend_u8(np.full_like(img, mean), img, fc)
        elif op == 2:
            img = _blend_u8(np.repeat(_gray_u8(img)[..., None], 3, -1), img, fs)
        else:
            h, s_, v = _rgb2hsv_u8(img)
            shift = int(fh * 255) & 255                                     # np.uint8 cast of hue_factor * 255: truncation, modulo 256
            img = _hsv2rgb_u8((h.astype(np.int64) + shift).astype(np.uint8), s_, v)
    return img


def project_labels(coarse_points, P, K_2, K_4, img_H, img_W, num_kpt, rs, nuscenes=False):
    """kitti.py:334-372 (nuscenes.py:254-296 with nuscenes=True: too few in-picture points -> zero indices and valid_kpt False instead of
    a short draw; no assertion on the fine pixels): coarse (1/8) and fine (1/2) correspondences of the coarsest-stage points.
    coarse_points (3, n) float32."""
    scale_size = 0.125
    Rinv = np.linalg.inv(P[0:3, 0:3])
    proj = np.dot(K_4, np.dot(Rinv, coarse_points) - np.dot(Rinv, P[0:3, 3:]))
    mask = np.zeros((1, coarse_points.shape[1]), dtype=np.float32)
    proj[0:2, :] = proj[0:2, :] / proj[2:, :]
    xy = np.floor(proj[0:2, :] + 0.5)
    inpic = (xy[0] >= 1) & (xy[0] <= (img_W * scale_size - 3)) & (xy[1] >= 1) & (xy[1] <= (img_H * scale_size - 3)) & (proj[2] > 0)
    mask[:, inpic] = 1.0
    pc_kpt_idx = np.where(mask.squeeze() == 1)[0]
    valid_kpt = True
    if not nuscenes or len(pc_kpt_idx) >= num_kpt:
        pc_kpt_idx = pc_kpt_idx[rs.permutation(len(pc_kpt_idx))[0:num_kpt]]
    else:
        valid_kpt = False
        pc_kpt_idx = np.zeros((num_kpt,), dtype=np.int64)
    pc_outline_idx = np.where(mask.squeeze() == 0)[0]
    pc_outline_idx = pc_outline_idx[rs.permutation(len(pc_outline_idx))[0:num_kpt]]
    xy2 = xy[:, inpic]
    H8, W8 = int(img_H * scale_size), int(img_W * scale_size)
    img_mask_s8 = np.zeros((H8, W8))
    np.add.at(img_mask_s8, (xy2[1].astype(np.int64), xy2[0].astype(np.int64)), 1.0)   # coo_matrix(...).toarray() sums duplicates
    img_mask_s8[img_mask_s8 > 0] = 1.0
    coarse_xy = xy[:, pc_kpt_idx]
    img_kpt_s8_index = xy[1, pc_kpt_idx] * img_W * scale_size + xy[0, pc_kpt_idx]
    img_outline_index = np.where(img_mask_s8.reshape(-1) == 0)[0]
    img_outline_index = img_outline_index[rs.permutation(len(img_outline_index))[0:num_kpt]]
    kpts = coarse_points[:, pc_kpt_idx]
    pp = np.dot(K_2, np.dot(Rinv, kpts) - np.dot(Rinv, P[0:3, 3:]))
    pp[0:2, :] = pp[0:2, :] / pp[2:, :]
    fine_xy = np.floor(pp[0:2, :])
    fine_in = (fine_xy[0] >= 0) & (fine_xy[0] <= (img_W * 0.5 - 1)) & (fine_xy[1] >= 0) & (fine_xy[1] <= (img_H * 0.5 - 1)) & (pp[2] > 0)
    assert nuscenes or np.all(fine_in)
    return {
        **({"valid_kpt": valid_kpt} if nuscenes else {}),
        "coarse_img_mask": img_mask_s8.astype(np.float32),
        "pc_kpt_idx": pc_kpt_idx,
        "pc_outline_idx": pc_outline_idx,
        "fine_xy_coors": fine_xy.astype(np.int32),
        "coarse_img_kpt_idx": img_kpt_s8_index.astype(np.int64),
        "fine_img_kpt_index": (fine_xy[1, :] * img_W * 0.5 + fine_xy[0, :]).astype(np.int64),
        "fine_center_kpt_coors": (coarse_xy * 4).astype(np.int32),
        "coarse_img_outline_index": img_outline_index.astype(np.int64),
    }


def point2node(nodes, points):
    """model/network.py:250-264: nearest node (expansion-form squared distance, float32) of every point; (M,3), (n,3) -> (n,)."""
    nodes = nodes.astype(np.float32)
    points = points.astype(np.float32)
    d = -2 * (points @ nodes.T)
    d = d + (points ** 2).sum(-1)[:, None]
    d = d + (nodes ** 2).sum(-1)[None, :]
    return np.argmin(d, axis=1)


def prepare_frame(data, img, K, P_Tr, index, opt, mode="val"):
    """The whole of kitti.py:259-393 (val mode; mode='train': random crop + colour jitter) minus the disk reads and the KNN tables (oracle/cofi_oracle.build_pyramid covers
    those).  `opt` carries img_H, img_W, num_pc, num_kpt and the six P_*_amplitude values.  Returns the reference's dict plus the
    intermediates the tests compare stage by stage."""
    seed = frame_seed(index)
    rs = np.random.RandomState(seed)   # == the global numpy state after np.random.seed(seed)
    rnd = random.Random(seed)          # == the global `random` after random.seed(seed)
    intensity, sn, pc = data[3:4, :], data[4:, :], data[0:3, :]
    pc = rigid(P_Tr, pc)
    sn = rigid(P_Tr, sn, translate=False)
    vpc, vint, vsn = voxel_down_sample(pc, intensity, sn, 0.1)
    choice = downsample_choice(vpc.shape[1], opt.num_pc, rs)
    pc, intensity, sn = vpc[:, choice], vint[:, choice], vsn[:, choice]
    amp = (opt.P_tx_amplitude, opt.P_ty_amplitude, opt.P_tz_amplitude, opt.P_Rx_amplitude, opt.P_Ry_amplitude, opt.P_Rz_amplitude)
    P = random_transform(rnd, amp)
    pc = rigid(P, pc)
    sn = rigid(P, sn, translate=False)
    # preprocess_data.py:55-59: random half sub-sampling WITH replacement, stage i + 1 from stage i
    sub, n = [], pc.shape[1]
    for _ in range(NUM_STAGES - 1):
        sub.append(rs.choice(np.arange(n), size=n // 2))
        n //= 2
    points = [np.ascontiguousarray(pc.T)]
    for s in sub:
        points.append(points[-1][s])
    feats = np.concatenate([intensity, sn], axis=0).T.astype(np.float32)
    small = resize_linear_u8(img, int(round(img.shape[1] * 0.5)), int(round(img.shape[0] * 0.5)))
    K = camera_matrix_scaling(K, 0.5)
    if mode == "train":
        dx = rnd.randint(0, small.shape[1] - opt.img_W)
        dy = rnd.randint(0, small.shape[0] - opt.img_H)
    else:
        dx = int((small.shape[1] - opt.img_W) / 2)
        dy = int((small.shape[0] - opt.img_H) / 2)
    crop = small[dy:dy + opt.img_H, dx:dx + opt.img_W, :]
    K = camera_matrix_cropping(K, dx=dx, dy=dy)
    K_2 = camera_matrix_scaling(K, 0.5)
    K_4 = camera_matrix_scaling(K, 0.125)
    if mode == "train":
        crop = color_jitter(crop, *jitter_params(seed))                     # kitti.py:329-330
    coarse_points = np.array(points[-1], dtype=np.float32).T
    out = project_labels(coarse_points, P, K_2, K_4, opt.img_H, opt.img_W, opt.num_kpt, rs)
    out["fine_pc_inline_index"] = point2node(points[1], points[-1][out["pc_kpt_idx"]])
    out.update({
        "img": np.ascontiguousarray((crop.astype(np.float32) / 255.0).transpose(2, 0, 1)),
        "K": K_2.astype(np.float32), "K_4": K_4.astype(np.float32), "P": np.linalg.inv(P).astype(np.float32), "index": index,
        # intermediates
        "points": points, "feats": feats, "subsample": sub, "choice": choice, "P_random": P,
        "voxel": (vpc, vint, vsn), "crop": (dy, dx), "resized_hw": small.shape[:2],
    })
    return out


def prepare_frame_nuscenes(pc4, img, K, index, opt, mode="val"):
    """nuscenes.py:177-320 (val mode; mode='train': random crop + colour jitter, nuscenes.py:232-234, 249-250) minus the disk reads and the KNN tables: seed = index, the stored (4, N) cloud [xyz | intensity]
    is already in the camera frame and is resampled directly (the voxel grid is commented out there), features = [intensity | point]."""
    seed = int(index)
    rs = np.random.RandomState(seed)
    rnd = random.Random(seed)
    intensity, pc = pc4[3, :].reshape(1, -1), pc4[0:3, :]
    choice = downsample_choice(pc.shape[1], opt.num_pc, rs)
    pc, intensity = pc[:, choice], intensity[:, choice]
    amp = (opt.P_tx_amplitude, opt.P_ty_amplitude, opt.P_tz_amplitude, opt.P_Rx_amplitude, opt.P_Ry_amplitude, opt.P_Rz_amplitude)
    P = random_transform(rnd, amp)
    pc = rigid(P, pc)
    sub, n = [], pc.shape[1]
    for _ in range(NUM_STAGES - 1):
        sub.append(rs.choice(np.arange(n), size=n // 2))
        n //= 2
    points = [np.ascontiguousarray(pc.T)]
    for s_ in sub:
        points.append(points[-1][s_])
    feats = np.concatenate([intensity, pc], axis=0).T.astype(np.float32)
    small = resize_linear_u8(img, int(round(img.shape[1] * 0.5)), int(round(img.shape[0] * 0.5)))
    K = camera_matrix_scaling(K, 0.5)
    if mode == "train":
        dx = rnd.randint(0, small.shape[1] - opt.img_W)
        dy = rnd.randint(0, small.shape[0] - opt.img_H)
    else:
        dx = int((small.shape[1] - opt.img_W) / 2)
        dy = int((small.shape[0] - opt.img_H) / 2)
    crop = small[dy:dy + opt.img_H, dx:dx + opt.img_W, :]
    K = camera_matrix_cropping(K, dx=dx, dy=dy)
    K_2 = camera_matrix_scaling(K, 0.5)
    K_4 = camera_matrix_scaling(K, 0.125)
    if mode == "train":
        crop = color_jitter(crop, *jitter_params(seed))                     # nuscenes.py:249-250
    coarse_points = np.array(points[-1], dtype=np.float32).T
    out = project_labels(coarse_points, P, K_2, K_4, opt.img_H, opt.img_W, opt.num_kpt, rs, nuscenes=True)
    out["fine_pc_inline_index"] = point2node(points[1], points[-1][out["pc_kpt_idx"]])
    out.update({
        "img": np.ascontiguousarray((crop.astype(np.float32) / 255.0).transpose(2, 0, 1)),
        "K": K_2.astype(np.float32), "K_4": K_4.astype(np.float32), "P": np.linalg.inv(P).astype(np.float32),
        "points": points, "feats": feats, "subsample": sub, "choice": choice, "P_random": P, "crop": (dy, dx),
    })
    return out
