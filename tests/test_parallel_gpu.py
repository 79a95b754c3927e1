"""The RCCL ("nccl" backend) side of the frame-parallel path on real GPUs (BASELINE configs[3]):
  * one rank: the process group is RCCL, the result gather and the timing all-gather of bench.py run on device tensors - this is
    the first place the nccl backend executes at all (a 1-GPU box cannot host two RCCL ranks: RCCL refuses a shared device);
  * two ranks over RCCL / xGMI, the driver's own launch line for `bench.py --gpus 2`, whenever the box exposes >= 2 GPUs (skipped
    otherwise - the multi-GPU scaling run is the driver's).
Needs a real MI355X:  python -m pytest tests -m gpu"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RANK_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from cofii2p_amd.parallel import gather_frame_results, shard_frames
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", rank=rank, world_size=world)
assert dist.get_backend() == "nccl"
total = 7
ids = shard_frames(list(range(total)), rank, world)
vals = torch.tensor([[2.0 * i, i + 0.5, float(rank)] for i in ids], dtype=torch.float32, device=dev).reshape(len(ids), 3)
out = gather_frame_results(ids, vals, total)
expect = torch.tensor([[2.0 * i, i + 0.5, float(i %% world)] for i in range(total)], device=dev)
assert out.is_cuda and torch.equal(out, expect), out
t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == float(world)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", rank, world)
'''


def _torchrun(nproc, script_args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_rccl_process_group_and_result_gather_one_rank(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % ROOT)
    out = _torchrun(1, [str(script)])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL_OK 0 1" in out.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the node (the 8-GPU scaling run is the driver's)")
def test_rccl_two_ranks_gather_and_bench(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % ROOT)
    out = _torchrun(2, [str(script)])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL_OK 0 2" in out.stdout and "RCCL_OK 1 2" in out.stdout
    # the driver's launch line for N = 2: one rank per GPU over RCCL
    out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--points", "4096", "--distinct-frames", "4",
                        "--repeats", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-batch-sweep"], timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["dist_backend"] == "nccl" and d["gathered_frame_results"] == 8 and len(d["per_rank_frames_per_s"]) == 2
