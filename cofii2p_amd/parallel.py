"""Frame-level data parallelism: one process per GPU, frame i -> rank i mod W, weights
replicated, NO data-path collective.  The only exchange is the end-of-run gather of per-frame results
(a few floats per frame) over torch.distributed — RCCL over xGMI on the GPU node ("nccl" backend),
gloo in the CPU tests."""
from typing import List, Sequence

import torch


def shard_frames(frame_ids: Sequence[int], rank: int, world: int) -> List[int]:
    """Round-robin shard: rank r owns frames r, r+W, r+2W, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return [f for i, f in enumerate(frame_ids) if i % world == rank]


def gather_frame_results(frame_ids: Sequence[int], values: torch.Tensor, total_frames: int) -> torch.Tensor:
    """values (n_local, D) per-frame results of this rank (e.g. RRE/RTE, match counts), frame_ids the
    global ids they belong to.  Returns (total_frames, D) on every rank, rows in global frame order.
    Ragged shards are padded to the largest shard; the ids travel in their own int64 tensor (a float payload would round ids
    above 2^24, or above 2048 in half precision): two all_gathers per call."""
    import torch.distributed as dist

    ids = torch.as_tensor(list(frame_ids), dtype=torch.int64, device=values.device)
    if ids.numel() != values.shape[0]:
        raise ValueError("%d frame ids for %d result rows" % (ids.numel(), values.shape[0]))
    if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= total_frames):
        raise ValueError("frame id outside [0, %d)" % total_frames)
    if not (dist.is_available() and dist.is_initialized()):
        out = torch.zeros((total_frames, values.shape[1]), dtype=values.dtype, device=values.device)
        out[ids] = values
        return out
    world = dist.get_world_size()
    n_max = (total_frames + world - 1) // world
    D = values.shape[1]
    n = ids.numel()
    if n > n_max:
        raise ValueError("a rank holds %d frames, more than ceil(%d / %d)" % (n, total_frames, world))
    pad_ids = torch.full((n_max,), -1, dtype=torch.int64, device=values.device)
    pad_val = torch.zeros((n_max, D), dtype=values.dtype, device=values.device)
    pad_ids[:n] = ids
    pad_val[:n] = values
    b_ids = [torch.empty_like(pad_ids) for _ in range(world)]
    b_val = [torch.empty_like(pad_val) for _ in range(world)]
    dist.all_gather(b_ids, pad_ids)
    dist.all_gather(b_val, pad_val)
    out = torch.zeros((total_frames, D), dtype=values.dtype, device=values.device)
    for i_, v_ in zip(b_ids, b_val):
        ok = i_ >= 0
        out[i_[ok]] = v_[ok]
    return out
