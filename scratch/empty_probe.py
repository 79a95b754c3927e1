import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cofii2p_amd import ops
dev = "cuda:0"
def tryit(name, fn):
    try:
        r = fn(); torch.cuda.synchronize()
        print("OK  ", name, tuple(r.shape) if torch.is_tensor(r) else [tuple(t.shape) for t in r if torch.is_tensor(t)])
    except Exception as e:
        print("EXC ", name, type(e).__name__, str(e)[:120])
z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
r = lambda *s: torch.randn(*s, device=dev)
tryit("knn Q=0", lambda: ops.knn(r(100, 3), z(0, 3), 16))
tryit("knn grid Q=0", lambda: ops.knn(sup := r(2000, 3), z(0, 3), 16, grid=ops.KnnGrid(sup)))
tryit("knn S=0", lambda: ops.knn(z(0, 3), r(5, 3), 16))
tryit("gemm M=0", lambda: ops.gemm(z(0, 64), r(32, 64)))
tryit("gemm_colstats M=0", lambda: ops.gemm_colstats(z(0, 64), r(32, 64)))
tryit("gather_rows 0", lambda: ops.gather_rows(r(10, 8), z(0, dt=torch.int32)))
tryit("l2norm_rows 0", lambda: ops.l2norm_rows(z(0, 64)))
tryit("layer_norm 0", lambda: ops.layer_norm(z(0, 64), r(64), r(64)))
tryit("neighbor_maxpool M=0", lambda: ops.neighbor_maxpool(r(10, 32), z(0, 128, dt=torch.int32)))
tryit("attention L=0", lambda: ops.attention(z(0, 128), r(16, 128), r(16, 128), 4))
tryit("nearest_node Q=0", lambda: ops.nearest_node(r(10, 3), z(0, 3)))
cnt = torch.zeros(2, dtype=torch.int32, device=dev)
tryit("fine_match n=0", lambda: ops.fine_match(z(8, 64, 16), z(8, 64), z(2, 8), cnt, 1.0))
tryit("extract_patches n=0", lambda: ops.extract_patches_nhwc(r(80 * 256, 64), 80, 256, z(2, 8), cnt, 8, 4.0))
