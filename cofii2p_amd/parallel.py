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
    Ragged shards are padded to the largest shard; one all_gather per call."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        out = torch.zeros((total_frames, values.shape[1]), dtype=values.dtype, device=values.device)
        out[torch.as_tensor(list(frame_ids), dtype=torch.long, device=values.device)] = values
        return out
    world = dist.get_world_size()
    n_max = (total_frames + world - 1) // world
    D = values.shape[1]
    pad = torch.full((n_max, D + 1), -1.0, dtype=values.dtype, device=values.device)
    n = len(frame_ids)
    if n:
        pad[:n, 0] = torch.as_tensor(list(frame_ids), dtype=values.dtype, device=values.device)
        pad[:n, 1:] = values
    bucket = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bucket, pad)
    out = torch.zeros((total_frames, D), dtype=values.dtype, device=values.device)
    for b in bucket:
        ok = b[:, 0] >= 0
        out[b[ok, 0].long()] = b[ok, 1:]
    return out
