"""The random draws of one dataset __getitem__(index) call in the reference's order (data/kitti.py:261-264,170-176,220-229,
model/kpconv/preprocess_data.py:58, data/kitti.py:313-314,345-358) - numpy + stdlib only, NO torch import: the pipelined loader
(cofii2p_amd/loader.py) runs `draw_frame` in spawned worker processes, which must start fast and never touch HIP."""
import random
from typing import Optional

import numpy as np

NUM_STAGES = 5


def frame_seed(index: int) -> int:
    """kitti.py:261-262"""
    (seed,) = np.random.SeedSequence([index]).generate_state(1)
    return int(seed)


class FrameSampler:
    """The random draws of one __getitem__(index) call, in the reference's order."""

    def __init__(self, index: int, seed: Optional[int] = None, dataset: str = "kitti"):
        # kitti.py:261-264 seeds with SeedSequence([index]); nuscenes.py:178-181 with the index itself
        self.seed = (frame_seed(index) if dataset == "kitti" else int(index)) if seed is None else int(seed)
        self.rs = np.random.RandomState(self.seed)   # the global numpy state after np.random.seed(seed)
        self.rnd = random.Random(self.seed)          # the global `random` state after random.seed(seed)

    def downsample_choice(self, n: int, num_pc: int) -> np.ndarray:
        """kitti.py:168-176"""
        if n >= num_pc:
            return self.rs.choice(n, num_pc, replace=False)
        fix = np.arange(n)
        while n + fix.shape[0] < num_pc:
            fix = np.concatenate((fix, np.arange(n)), axis=0)
        return np.concatenate((fix, self.rs.choice(n, num_pc - fix.shape[0], replace=False)), axis=0)

    def random_transform(self, opt) -> np.ndarray:
        """kitti.py:216-235 (+ :203-214): t, then the angles; R = Rz Ry Rx; 4x4 float32."""
        u = self.rnd.uniform
        t = [u(-opt.P_tx_amplitude, opt.P_tx_amplitude), u(-opt.P_ty_amplitude, opt.P_ty_amplitude), u(-opt.P_tz_amplitude, opt.P_tz_amplitude)]
        a = [u(-opt.P_Rx_amplitude, opt.P_Rx_amplitude), u(-opt.P_Ry_amplitude, opt.P_Ry_amplitude), u(-opt.P_Rz_amplitude, opt.P_Rz_amplitude)]
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        P = np.identity(4, dtype=np.float32)
        P[0:3, 0:3] = np.dot(Rz, np.dot(Ry, Rx))
        P[0:3, 3] = t
        return P

    def subsample_indices(self, n: int, num_stages: int = NUM_STAGES):
        """preprocess_data.py:55-59: half of the previous stage, WITH replacement."""
        out = []
        for _ in range(num_stages - 1):
            out.append(self.rs.choice(np.arange(n), size=n // 2))
            n //= 2
        return out

    def crop_offsets(self, small_hw, opt, mode: str):
        """kitti.py:312-317: random crop in train mode, centred otherwise."""
        h, w = small_hw
        if mode == "train":
            dx = self.rnd.randint(0, w - opt.img_W)
            dy = self.rnd.randint(0, h - opt.img_H)
        else:
            dx = int((w - opt.img_W) / 2)
            dy = int((h - opt.img_H) / 2)
        return dy, dx

    def permutation(self, n: int) -> np.ndarray:
        return self.rs.permutation(n)

    def color_jitter_params(self):
        """train mode (kitti.py:193-201): order of the four ColorJitter operations (0 brightness, 1 contrast, 2 saturation, 3 hue) and their
        factors, U(0.8, 1.2) x 3 and U(-0.1, 0.1).  torchvision draws them from the unseeded torch generator; here they come from the frame
        seed (a generator of their own: the reference's numpy / random streams are not disturbed), so a frame is reproducible."""
        rs = np.random.RandomState((int(self.seed) + 0x9E3779B9) % (1 << 32))
        order = [int(v) for v in rs.permutation(4)]
        fb, fc, fs = (float(v) for v in rs.uniform(0.8, 1.2, 3))
        return order, fb, fc, fs, float(rs.uniform(-0.1, 0.1))


def draw_frame(index: int, nvox: int, num_pc: int, amplitudes, dataset: str = "kitti", num_stages: int = NUM_STAGES):
    """Everything `FramePreparer.complete` draws before the KNN pyramid, for a worker process: -> dict with
    choice (num_pc,) int32, P (4,4) float32, sub [int32 arrays], rs_state / rnd_state (the generators AFTER these draws: the label
    permutations of project_labels continue from them).  amplitudes = (tx, ty, tz, Rx, Ry, Rz) of the options object."""
    class _Opt:
        P_tx_amplitude, P_ty_amplitude, P_tz_amplitude, P_Rx_amplitude, P_Ry_amplitude, P_Rz_amplitude = amplitudes

    s = FrameSampler(index, dataset=dataset)
    choice = s.downsample_choice(nvox, num_pc)
    P = s.random_transform(_Opt)
    sub = s.subsample_indices(num_pc, num_stages)
    return {"index": index, "nvox": nvox, "choice": np.ascontiguousarray(choice, dtype=np.int32), "P": np.ascontiguousarray(P, dtype=np.float32),
            "sub": [np.ascontiguousarray(x, dtype=np.int32) for x in sub], "rs_state": s.rs.get_state(), "rnd_state": s.rnd.getstate(), "seed": s.seed}


def sampler_from_state(d) -> "FrameSampler":
    """the FrameSampler a worker's draw_frame left behind (for the label permutations on the main process)"""
    s = FrameSampler.__new__(FrameSampler)
    s.seed = d["seed"]
    s.rs = np.random.RandomState()
    s.rs.set_state(d["rs_state"])
    s.rnd = random.Random()
    s.rnd.setstate(d["rnd_state"])
    return s


def worker_init():
    """initializer of the loader's worker processes (lives here so that unpickling it imports numpy only)"""


def worker_warm(seconds: float) -> int:
    import os
    import time

    time.sleep(seconds)
    return os.getpid()
