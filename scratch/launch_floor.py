"""Calibration: per-kernel cost of a dependent chain of trivial kernels inside a hipGraph (MI355X)."""
import torch, time
dev = torch.device("cuda", 0)
def chain(fn, n=500, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)
if __name__ == "__main__":
    t1 = torch.zeros(1, device=dev)
    print("1-element add_        : %.2f us/kernel" % chain(lambda: t1.add_(1.0)))
    t2 = torch.zeros(1 << 16, device=dev)
    print("64K-element add_      : %.2f us/kernel" % chain(lambda: t2.add_(1.0)))
    t3 = torch.zeros(1 << 20, device=dev)
    print("1M-element add_ (8MB) : %.2f us/kernel" % chain(lambda: t3.add_(1.0)))
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cofii2p_amd import ops
    ops.GEMM_MODE = "bf16x3"
    for (M, N, K) in [(1280, 128, 128), (1280, 256, 128), (1280, 128, 512), (320, 256, 2304), (5120, 64, 576)]:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); o = torch.empty(M, N, device=dev)
        print("gemm %5d %4d %5d   : %.2f us/call" % (M, N, K, chain(lambda: ops.gemm(a, w, out=o), n=100)))
        print("gemm_colstats          : %.2f us/call" % chain(lambda: ops.gemm_colstats(a, w, out=o), n=100))
