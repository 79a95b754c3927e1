"""Is the batch-1 loop CPU-bound?  Time spent enqueueing (forward_async) vs waiting (finish) per frame."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cofii2p_amd import ops
from cofii2p_amd.network import CoFiI2P
ops.GEMM_MODE = "bf16x3"
dev = torch.device("cuda", 0)
model = CoFiI2P(bench.Opt()).to(dev); model.enable_graphs()
frames = bench.make_inputs(dev, [0, 1, 2, 3], 20480)
S = 2
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
pend = [None] * S
def run(n, acc):
    for i in range(n):
        sl = i % S
        if pend[sl] is not None:
            t = time.perf_counter(); model.finish(pend[sl]); acc[1] += time.perf_counter() - t
        pyr, img, _ = frames[i % 4]
        t = time.perf_counter()
        with torch.cuda.stream(streams[sl]):
            pend[sl] = model.forward_async(sl, pyr, img)
        acc[0] += time.perf_counter() - t
    for sl in range(S):
        if pend[sl] is not None:
            model.finish(pend[sl]); pend[sl] = None
run(10, [0, 0]); torch.cuda.synchronize()
acc = [0.0, 0.0]; t0 = time.perf_counter(); run(100, acc); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("per frame: total %.3f ms | forward_async (CPU enqueue) %.3f ms | finish (wait + slicing) %.3f ms" % (10 * dt, 10 * acc[0], 10 * acc[1]))
# pure enqueue cost with the GPU idle: replay graph only
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50):
    h = model.forward_async(0, frames[0][0], frames[0][1]); 
    torch.cuda.synchronize()
print("serial latency per frame (enqueue + run + sync): %.3f ms" % (20 * (time.perf_counter() - t0)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run(40, [0, 0]); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
