"""Per-K-tile cost of the 64x64 bf16x3 GEMM kernel at small grids (forced plans)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cofii2p_amd import ops, _lib
from scratch.launch_floor import chain
ops.GEMM_MODE = "bf16x3"
lib = _lib.load()
dev = torch.device("cuda", 0)
for (M, N) in [(320, 256), (5120, 64), (1280, 128)]:
    for ks in (1, 4):
        row = []
        for K in (128, 256, 512, 1024, 2048, 4096):
            if K // ks < 128: continue
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); o = torch.empty(M, N, device=dev)
            lib.cofi_gemm_debug_force_plan(64, 64, ks)
            row.append("K%d: %.1f" % (K, chain(lambda: ops.gemm(a, w, out=o), n=50)))
        lib.cofi_gemm_debug_force_plan(0, 0, 0)
        print("M %d N %d ks %d  " % (M, N, ks) + "  ".join(row))
