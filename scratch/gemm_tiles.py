"""Is the big-shape bf16x3 GEMM bound by operand traffic per CU?  Same problem under the three tile shapes (forced plans)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cofii2p_amd import ops, _lib
from scratch.launch_floor import chain
ops.GEMM_MODE = "bf16x3"
lib = _lib.load()
dev = torch.device("cuda", 0)
for (M, N, K) in [(40960, 1024, 3072), (20480, 512, 7680), (81920, 512, 1536), (163840, 64, 960)]:
    a = torch.randn(M, K, device=dev); w = ops.presplit(torch.randn(N, K, device=dev)); o = torch.empty(M, N, device=dev)
    row = []
    for bm, bn in ((128, 128), (64, 128), (64, 64)):
        lib.cofi_gemm_debug_force_plan(bm, bn, 1)
        t = chain(lambda: ops.gemm(a, w, out=o), n=5, reps=3)
        row.append("%dx%d: %.0f us (%.0f TF/s, %.1f operand bytes/flop*1e3)" % (bm, bn, t, 2.0 * M * N * K / t * 1e-6, 1e3 * 4.0 * (bm + bn) / (2.0 * bm * bn)))
    lib.cofi_gemm_debug_force_plan(0, 0, 0)
    print("M %d N %d K %d  " % (M, N, K) + " | ".join(row))
