"""KNN pyramid build time per frame: brute force vs cell grids (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cofii2p_amd import ops, preprocess
from cofii2p_amd.synth import make_frame, subsample_indices

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
p0 = torch.from_numpy(make_frame(3, N).points).to(dev)
sub = [torch.from_numpy(s).to(dev) for s in subsample_indices(N, 5, seed=3)]

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for name, thr in (("grid", 4096), ("grid>=2048", 2048), ("grid>=1024", 1024), ("brute", 1 << 30)):
    ops.KNN_GRID_MIN_SUPPORT = thr
    print("%-12s pyramid %.3f ms/frame" % (name, timed(lambda: preprocess.build_pyramid(p0, sub))))
pts = [p0]
for s in sub: pts.append(pts[-1][s.long()].contiguous())
for i in (0, 1, 2):
    S = pts[i]
    g = ops.KnnGrid(S)
    print("stage %d S=%d: build %.3f ms, self grid %.3f (cell order %.3f) vs brute %.3f ms" % (
        i, S.shape[0], timed(lambda: ops.KnnGrid(S)), timed(lambda: ops.knn(S, S, 128, grid=g)),
        timed(lambda: ops.knn(S, S, 128, grid=g, qorder=g.order)), timed(lambda: ops.knn(S, S, 128))))
print("per search (ms): grid | brute")
tot_g = tot_b = 0.0
GMIN = int(os.environ.get("GMIN", "4096"))
grids = [ops.KnnGrid(p) if p.shape[0] >= GMIN else None for p in pts]
for i in range(len(pts)):
    todo = [("self%d" % i, pts[i], pts[i], grids[i])]
    if i < len(pts) - 1:
        todo += [("sub%d" % i, pts[i], pts[i + 1], grids[i]), ("up%d" % i, pts[i + 1], pts[i], grids[i + 1])]
    for name, S, Qr, g in todo:
        tb = timed(lambda: ops.knn(S, Qr, 128))
        tg = timed(lambda: ops.knn(S, Qr, 128, grid=g)) if g is not None else tb
        tot_g += tg; tot_b += tb
        print("  %-6s S=%5d Q=%5d  %.3f | %.3f" % (name, S.shape[0], Qr.shape[0], tg, tb))
print("sum: grid %.3f ms, brute %.3f ms (+ %d grid builds)" % (tot_g, tot_b, sum(g is not None for g in grids)))
