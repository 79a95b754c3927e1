"""The data side of one KITTI frame on the device: what kitti_pc_img_dataset.__getitem__
(data/kitti.py:259-393) does between the disk read and the model call, with the heavy parts as HIP kernels
(csrc/dataside.hip, csrc/knn_grid.hip):

    raw scan (7, N)  --calibration transform-->  0.1 m voxel grid  --resample to num_pc-->  random SE(3)  -->  KNN pyramid
    raw image (H, W, 3) uint8  --x0.5 bilinear resize, crop (train mode: random), [train mode: colour jitter], / 255, CHW-->  model image
    coarsest-stage points  --project with K/8 and K/2-->  coarse / fine labels                     (host numpy, 1280 points)

The reference seeds the GLOBAL numpy and `random` generators per frame index and draws from them in a fixed order
(kitti.py:261-264, then :170/:175, :220-229, preprocess_data.py:58 four times, :313-314 in train mode, :345/:349/:358);
`FrameSampler` (cofii2p_amd/sampler.py) makes the same draws in the same order from private generators, so a frame prepared here
gets the reference's choice INDICES, SE(3) and label permutations.  Caveat: the voxel table those indices select from is in ascending
voxel-index order here, in open3d's hash-map order in the reference - the same index therefore names a different voxel and the
resampled cloud equals the reference's in distribution, not point for point (exactly equal whenever the voxel table is, e.g. the
nuScenes path, which has no voxel grid).  The draws are host-side by nature (Mersenne Twister streams); they are handed to the
kernels as plain index arrays / a 4x4 matrix.  mode='train' (kitti.py:312-314, 329-330): the crop offsets come from the same `random`
stream as in the reference; torchvision's ColorJitter (brightness / contrast / saturation in (0.8, 1.2), hue in (-0.1, 0.1), random
order) runs as a kernel that is bit-equal to the PIL operations torchvision applies - its order and factors are drawn from the frame
seed (torchvision uses the unseeded torch generator), so a training frame is reproducible.  `prepare()` syncs once per frame on the voxel count (the resampling draw depends on
it); cofii2p_amd/loader.py pipelines frames so that nothing waits.
"""
import random
from typing import Dict, Optional

import ctypes

import numpy as np
import torch

from . import _lib, ops
from .preprocess import build_pyramid

VOXEL_SIZE = 0.1   # kitti.py:283


from .sampler import NUM_STAGES, FrameSampler, frame_seed  # noqa: E402,F401  (torch-free module: the loader's worker processes import it alone)


def calib_matrices(lines: Dict[str, str]) -> Dict[str, np.ndarray]:
    """KittiCalibHelper.read_calib_files for one sequence (kitti.py:24-63): {'Tr', 'P0'.., 'P0_K'..}; lines = {key: twelve numbers}."""
    out = {}
    for key, text in lines.items():
        mat = np.array(text.split(), dtype=np.float64).reshape(3, 4).astype(np.float32)
        P = np.identity(4, dtype=np.float32)
        if key == "Tr":
            P[0:3, :] = mat
        else:
            K = mat[0:3, 0:3]
            out[key + "_K"] = K
            tz = mat[2, 3]
            P[0:3, 3] = np.asarray([(mat[0, 3] - K[0, 2] * tz) / K[0, 0], (mat[1, 3] - K[1, 2] * tz) / K[1, 1], tz], dtype=np.float32)
        out[key] = P
    return out


def intrinsics_and_crop(K: np.ndarray, img_hw, opt, sampler: Optional[FrameSampler], mode: str = "val"):
    """kitti.py:306-328: size of the x0.5 image, crop offsets, K of the cropped image at 1/2 (-> 'K') and 1/8 (-> 'K_4') of it.
    Returns (K_2, K_4, (dy, dx), (resized_h, resized_w)) in K's own dtype (the reference keeps the dtype of the stored matrix)."""
    rh, rw = int(round(img_hw[0] * 0.5)), int(round(img_hw[1] * 0.5))

    def scaled(M, s):   # camera_matrix_scaling, kitti.py:188-191
        Ms = s * M
        Ms[2, 2] = 1
        return Ms

    Ks = scaled(K, 0.5)
    dy, dx = (sampler or FrameSampler(0)).crop_offsets((rh, rw), opt, mode) if mode == "train" else (int((rh - opt.img_H) / 2), int((rw - opt.img_W) / 2))
    Kc = np.copy(Ks)    # camera_matrix_cropping, kitti.py:182-186
    Kc[0, 2] -= dx
    Kc[1, 2] -= dy
    return scaled(Kc, 0.5), scaled(Kc, 0.125), (dy, dx), (rh, rw)


def project_labels(coarse_points: np.ndarray, P: np.ndarray, K_2: np.ndarray, K_4: np.ndarray, opt, sampler: FrameSampler,
                   dataset: str = "kitti") -> Dict:
    """kitti.py:334-372 on the coarsest-stage points (n, 3): projection to the 1/8 image, in-picture mask, `num_kpt` random in / out
    points, occupied-pixel mask, then the fine (1/2 image) pixel of every kept point.  Plain numpy on 1280 points, written the way
    the reference writes it (same operations, same dtypes): nothing here is worth a kernel launch."""
    cp = np.ascontiguousarray(coarse_points, dtype=np.float32).T          # (3, n)
    s8 = 0.125
    Rinv = np.linalg.inv(P[0:3, 0:3])
    proj = np.dot(K_4, np.dot(Rinv, cp) - np.dot(Rinv, P[0:3, 3:]))
    proj[0:2, :] = proj[0:2, :] / proj[2:, :]
    xy = np.floor(proj[0:2, :] + 0.5)
    inpic = (xy[0] >= 1) & (xy[0] <= (opt.img_W * s8 - 3)) & (xy[1] >= 1) & (xy[1] <= (opt.img_H * s8 - 3)) & (proj[2] > 0)
    pc_kpt_idx = np.where(inpic)[0]
    valid_kpt = True
    if dataset == "kitti" or len(pc_kpt_idx) >= opt.num_kpt:
        pc_kpt_idx = pc_kpt_idx[sampler.permutation(len(pc_kpt_idx))[0:opt.num_kpt]]
    else:   # nuscenes.py:262-268: too few points in the picture - no draw, all-zero indices, the sample is flagged
        valid_kpt = False
        pc_kpt_idx = np.zeros((opt.num_kpt,), dtype=np.int64)
    pc_outline_idx = np.where(~inpic)[0]
    pc_outline_idx = pc_outline_idx[sampler.permutation(len(pc_outline_idx))[0:opt.num_kpt]]
    H8, W8 = int(opt.img_H * s8), int(opt.img_W * s8)
    mask = np.zeros((H8, W8), dtype=np.float32)
    mask[xy[1, inpic].astype(np.int64), xy[0, inpic].astype(np.int64)] = 1.0
    coarse_xy = xy[:, pc_kpt_idx]
    img_outline = np.where(mask.reshape(-1) == 0)[0]
    img_outline = img_outline[sampler.permutation(len(img_outline))[0:opt.num_kpt]]
    pp = np.dot(K_2, np.dot(Rinv, cp[:, pc_kpt_idx]) - np.dot(Rinv, P[0:3, 3:]))
    pp[0:2, :] = pp[0:2, :] / pp[2:, :]
    fine_xy = np.floor(pp[0:2, :])
    ok = (fine_xy[0] >= 0) & (fine_xy[0] <= (opt.img_W * 0.5 - 1)) & (fine_xy[1] >= 0) & (fine_xy[1] <= (opt.img_H * 0.5 - 1)) & (pp[2] > 0)
    if dataset == "kitti" and not np.all(ok):
        raise AssertionError("a coarse in-picture point projects outside the 1/2 image (kitti.py:366)")
    extra = {} if dataset == "kitti" else {"valid_kpt": valid_kpt}
    return {
        **extra,
        "coarse_img_mask": mask,
        "pc_kpt_idx": pc_kpt_idx,
        "pc_outline_idx": pc_outline_idx,
        "fine_xy_coors": fine_xy.astype(np.int32),
        "coarse_img_kpt_idx": (xy[1, pc_kpt_idx] * opt.img_W * s8 + xy[0, pc_kpt_idx]).astype(np.int64),
        "fine_img_kpt_index": (fine_xy[1, :] * opt.img_W * 0.5 + fine_xy[0, :]).astype(np.int64),
        "fine_center_kpt_coors": (coarse_xy * 4).astype(np.int32),
        "coarse_img_outline_index": img_outline.astype(np.int64),
    }


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class FramePreparer:
    """Device-side __getitem__: `prepare(data, img, K, P_Tr, index)` returns the reference's sample dict (kitti.py:375-393) with
    the tensors on the device (`img` (3,H,W), `pc_data_dict` with int64 tables and `feats`, the label tensors, K / K_4 / P).
    Buffers are cached per input size; one preparer per stream."""

    def __init__(self, opt, device="cuda", mode: str = "val", dataset: str = "kitti"):
        """dataset: 'kitti' (data/kitti.py: calibration transform, voxel grid, normals as features) or 'nuscenes' (data/nuscenes.py:177-320:
        the stored cloud is already in the camera frame and is resampled directly, features = [intensity | point], seed = index)."""
        if mode not in ("val", "train"):
            raise ValueError("mode must be 'val' or 'train'")
        if dataset not in ("kitti", "nuscenes"):
            raise ValueError("dataset name invalid, only support KITTI Odometry and Nuscenes now!")   # train.py:130
        self.dataset = dataset
        self.opt, self.device, self.mode = opt, torch.device(device), mode
        self._ws = ops.Workspace()
        self._rows = self._vox = None
        self.last = {}

    def voxel_downsample(self, data: torch.Tensor, P_Tr: torch.Tensor, wait: bool = True):
        """(7, N) raw scan + 4x4 calibration transform (device) -> ((cap, 8) voxel rows, count).  wait=True syncs on the count;
        wait=False returns (rows, pinned host counters, event) - read the counters after event.synchronize()."""
        lib = _lib.load()
        if data.dim() != 2 or data.shape[0] != 7 or data.dtype != torch.float32 or not data.is_contiguous():
            raise _lib.CofiError("dataside: the raw scan must be a contiguous (7, N) float32 tensor")
        N = data.shape[1]
        if self._rows is None or self._rows.shape[0] < N:
            self._rows = torch.empty((N, 8), dtype=torch.float32, device=self.device)
            self._vox = torch.empty((N, 8), dtype=torch.float32, device=self.device)
        rows, vox = self._rows[:N], self._vox[:N]
        cnt = torch.empty(2, dtype=torch.int32, device=self.device)
        _lib.check(lib.cofi_pack_transform_scan(_p(data), N, _p(P_Tr), _p(rows), _stream()), "cofi_pack_transform_scan")
        nbytes = lib.cofi_voxel_downsample_workspace(N)
        ws = self._ws.get(nbytes, self.device)
        _lib.check(lib.cofi_voxel_downsample(_p(rows), N, VOXEL_SIZE, _p(vox), N, _p(cnt), _p(ws), ws.numel(), _stream()), "cofi_voxel_downsample")
        if not wait:
            if getattr(self, "_cnt_host", None) is None:
                self._cnt_host = torch.empty(2, dtype=torch.int32, pin_memory=True)
            self._cnt_host.copy_(cnt, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return vox, self._cnt_host, ev
        c = cnt.cpu()
        if int(c[1]):
            raise _lib.CofiError("dataside: the scan spans more than 8192 voxels along an axis")
        return vox, int(c[0])

    def resample_transform(self, vox: torch.Tensor, choice: np.ndarray, P: np.ndarray):
        """rows `choice` of the voxel table, x' = R x + t, n' = R n -> points (n, 3), feats (n, 4) = [intensity | n'] (kitti.py:284-288, 293)
        or [intensity | x'] (nuscenes.py:199-204)."""
        ch = torch.from_numpy(np.ascontiguousarray(choice, dtype=np.int32)).to(self.device, non_blocking=True)
        Pd = torch.from_numpy(np.ascontiguousarray(P, dtype=np.float32)).to(self.device, non_blocking=True)
        return self.resample_transform_dev(vox, ch, Pd)

    def resample_transform_dev(self, vox: torch.Tensor, ch: torch.Tensor, Pd: torch.Tensor):
        """the same with the draw already on the device (int32 indices, 4x4 float32): no host work, capturable in a hipGraph"""
        lib = _lib.load()
        n = int(ch.shape[0])
        points = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        feats = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        _lib.check(lib.cofi_gather_transform(_p(vox), _p(ch), n, _p(Pd), _p(points), _p(feats), int(self.dataset == "nuscenes"), _stream()),
                   "cofi_gather_transform")
        return points, feats

    def image(self, img_u8: torch.Tensor, resized_hw, crop_yx):
        """(H, W, 3) uint8 device tensor -> (3, img_H, img_W) float32 (kitti.py:306-322, 375)."""
        lib = _lib.load()
        if img_u8.dim() != 3 or img_u8.shape[2] != 3 or img_u8.dtype != torch.uint8 or not img_u8.is_contiguous():
            raise _lib.CofiError("dataside: the image must be a contiguous (H, W, 3) uint8 tensor")
        out = torch.empty((3, self.opt.img_H, self.opt.img_W), dtype=torch.float32, device=self.device)
        _lib.check(lib.cofi_resize_crop_image(_p(img_u8), img_u8.shape[0], img_u8.shape[1], resized_hw[0], resized_hw[1], crop_yx[0], crop_yx[1],
                                              self.opt.img_H, self.opt.img_W, _p(out), _stream()), "cofi_resize_crop_image")
        return out

    def color_jitter(self, image: torch.Tensor, order, brightness: float, contrast: float, saturation: float, hue: float):
        """train mode (kitti.py:193-201, 329-330): torchvision ColorJitter on the cropped image, in place on the (3, H, W) float image
        (bit-equal to the PIL operations torchvision applies: cofi_color_jitter_chw)."""
        lib = _lib.load()
        ws = torch.empty((1,), dtype=torch.int64, device=image.device)
        arr = (ctypes.c_int * 4)(*[int(v) for v in order])
        _lib.check(lib.cofi_color_jitter_chw(_p(image), image.shape[1], image.shape[2], arr, float(brightness), float(contrast), float(saturation),
                                             float(hue), _p(ws), 8, _stream()), "cofi_color_jitter_chw")
        return image

    def begin(self, data, img, K: np.ndarray, P_Tr: np.ndarray, index: int):
        """First half of `prepare`: uploads (if needed) and enqueues everything that does not depend on a random draw - calibration
        transform + voxel grid - WITHOUT waiting for the voxel count.  A loader that keeps several frames in flight calls begin() for a
        later frame right after submitting the current frame's forward on the same stream, and complete() when it comes back to it:
        the count has long arrived and the host never blocks on the GPU.  One outstanding begin() per preparer."""
        dev = self.device
        if getattr(self, "_outstanding", False):
            raise _lib.CofiError("FramePreparer.begin: the previous begin() of this preparer has not been complete()d - its voxel table and "
                                 "count buffer would be overwritten (one preparer per frame in flight)")
        self._outstanding = True
        data = (torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)) if isinstance(data, np.ndarray) else data).to(dev, non_blocking=True)
        img = (torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img).to(dev, non_blocking=True)
        h = {"img": img, "K": K, "index": index}
        if self.dataset == "kitti":
            Ptr = torch.from_numpy(np.ascontiguousarray(P_Tr, dtype=np.float32)).to(dev, non_blocking=True)
            h["vox"], h["cnt_host"], h["event"] = self.voxel_downsample(data, Ptr, wait=False)
        else:
            # nuscenes.py:189-197: (4, N) = [xyz | intensity], already in the camera frame, no voxel grid (commented out in the reference):
            # the same (N, 8) row layout through the pack kernel with the identity transform and zero normals
            if data.dim() != 2 or data.shape[0] != 4:
                raise _lib.CofiError("dataside: a nuScenes cloud is (4, N) = [xyz | intensity]")
            nvox = data.shape[1]
            d7 = torch.zeros((7, nvox), dtype=torch.float32, device=dev)
            d7[:4] = data
            vox = torch.empty((nvox, 8), dtype=torch.float32, device=dev)
            eye = torch.eye(4, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().cofi_pack_transform_scan(_p(d7), nvox, _p(eye), _p(vox), _stream()), "cofi_pack_transform_scan")
            h["vox"], h["nvox"] = vox, nvox
        return h

    def prepare(self, data, img, K: np.ndarray, P_Tr: np.ndarray, index: int, defer_labels: bool = False) -> Dict:
        """defer_labels=True: the model inputs come back at once (ONE host sync: the voxel count); the label tensors, which need the
        coarsest-stage points on the host, are produced by calling out["finish_labels"]() later - e.g. after the forward of this
        frame has been submitted, when the points have long arrived - and are then added to the same dict."""
        return self.complete(self.begin(data, img, K, P_Tr, index), defer_labels)

    def complete(self, h: Dict, defer_labels: bool = False) -> Dict:
        """Second half of `prepare` (see begin): the draws that need the voxel count, resampling + SE(3), KNN pyramid, image, labels."""
        opt, dev = self.opt, self.device
        self._outstanding = False
        img, K, index = h["img"], h["K"], h["index"]
        s = FrameSampler(index, dataset=self.dataset)
        vox = h["vox"]
        if "event" in h:
            h["event"].synchronize()
            if int(h["cnt_host"][1]):
                raise _lib.CofiError("dataside: the scan spans more than 8192 voxels along an axis")
            nvox = int(h["cnt_host"][0])
        else:
            nvox = h["nvox"]
        choice = s.downsample_choice(nvox, opt.num_pc)
        P = s.random_transform(opt)
        points, feats = self.resample_transform(vox, choice, P)
        sub = s.subsample_indices(opt.num_pc, NUM_STAGES)
        pyr = build_pyramid(points, [torch.from_numpy(i).to(dev, non_blocking=True) for i in sub], int64=True)
        pyr["feats"] = feats
        K_2, K_4, crop, rhw = intrinsics_and_crop(K, img.shape[:2], opt, s, self.mode)
        image = self.image(img, rhw, crop)
        if self.mode == "train":
            self.color_jitter(image, *s.color_jitter_params())
        self.last = {"voxels": nvox, "choice": choice, "P_random": P, "subsample": sub, "crop": crop}
        out = {"img": image, "pc_data_dict": pyr,
               "K": torch.from_numpy(K_2.astype(np.float32)).to(dev, non_blocking=True), "K_4": torch.from_numpy(K_4.astype(np.float32)).to(dev, non_blocking=True),
               "P": torch.from_numpy(np.linalg.inv(P).astype(np.float32)).to(dev, non_blocking=True), "index": index}
        coarse_host = torch.empty(pyr["points"][-1].shape, dtype=torch.float32, pin_memory=True)
        coarse_host.copy_(pyr["points"][-1], non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()

        def finish_labels():
            ready.synchronize()
            lab = project_labels(coarse_host.numpy(), P, K_2, K_4, opt, s, dataset=self.dataset)
            kpt = torch.from_numpy(lab["pc_kpt_idx"]).to(dev)
            out["fine_pc_inline_index"] = ops.nearest_node(pyr["points"][1], pyr["points"][-1][kpt].contiguous()).to(torch.int64)   # point2node, kitti.py:374
            for k, v in lab.items():
                out[k] = torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v
            out.pop("finish_labels", None)
            return out

        if defer_labels:
            out["finish_labels"] = finish_labels
            return out
        return finish_labels()
