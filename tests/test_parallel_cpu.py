"""world_size-2 gloo test of the frame sharding + result gather (the N>1 path of bench.py /
eval-style drivers); runs on CPU."""
import os
import socket

import torch
import torch.multiprocessing as mp

from cofii2p_amd.parallel import gather_frame_results, shard_frames


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard_frames(list(range(total)), rank, world)
    vals = torch.tensor([[float(i) * 2.0, float(i) + 0.5, float(rank)] for i in ids], dtype=torch.float32).reshape(len(ids), 3)
    out = gather_frame_results(ids, vals, total)
    q.put((rank, out.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition():
    for world in (1, 2, 3, 8):
        shards = [shard_frames(list(range(21)), r, world) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(21))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1


def test_gather_two_ranks_ragged():
    total, world = 7, 2  # ragged: 4 + 3 frames
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = torch.tensor([[i * 2.0, i + 0.5, float(i % world)] for i in range(total)])
    for r in range(world):
        assert torch.equal(got[r], expect)


def test_gather_without_process_group():
    out = gather_frame_results([2, 0], torch.tensor([[1.0], [3.0]]), 3)
    assert out.flatten().tolist() == [3.0, 0.0, 1.0]


def test_gather_keeps_large_frame_ids_exact():
    """ids travel as int64, not inside the (possibly half-precision) payload: 2^24 + 1 and a bf16 payload stay exact"""
    total = (1 << 24) + 8
    ids = [total - 1, (1 << 24) + 1, 5]
    vals = torch.tensor([[1.0], [2.0], [3.0]], dtype=torch.bfloat16)
    out = gather_frame_results(ids, vals, total)
    assert out.dtype == torch.bfloat16 and float(out[total - 1]) == 1.0 and float(out[(1 << 24) + 1]) == 2.0 and float(out[5]) == 3.0
    assert float(out[1 << 24]) == 0.0
    import pytest

    with pytest.raises(ValueError):
        gather_frame_results([total], torch.zeros(1, 1), total)


def test_loader_worker_pools_do_not_oversubscribe_a_node(monkeypatch):
    """eight ranks' FrameLoader pools (one process per GPU) are sized from the cores the rank can count on - its affinity mask and an equal
    share of the node among LOCAL_WORLD_SIZE ranks, one core kept for the launch thread - not from the requested count alone"""
    import os

    from cofii2p_amd.loader import FrameLoader

    wb = FrameLoader.worker_budget
    # 8 ranks on a 64-core node, 4 workers requested each: 8 x (4 + 1 launch thread) = 40 <= 64
    assert wb(4, affinity=range(64), local_world=8, online=64) == 4
    # ... on a 32-core node: the share of a rank is 4 cores -> 3 workers, 8 x (3 + 1) = 32
    assert wb(4, affinity=range(32), local_world=8, online=32) == 3
    # a rank pinned to 2 cores keeps one worker whatever was asked for
    assert wb(4, affinity=range(2), local_world=1, online=64) == 1
    for lw in (1, 2, 4, 8):
        for cores in (8, 16, 96, 192):
            w = wb(4, affinity=range(cores), local_world=lw, online=cores)
            assert 1 <= w <= 4 and (lw * (w + 1) <= cores or w == 1)
    # defaults come from the process itself: os.sched_getaffinity and LOCAL_WORLD_SIZE
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    n = len(os.sched_getaffinity(0))
    assert wb(4) == max(1, min(4, max(1, min(n, (os.cpu_count() or n) // 8)) - 1))
