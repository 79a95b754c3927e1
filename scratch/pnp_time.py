import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from cofii2p_amd import pose
from test_pose_cpu import K, synth
from scratch.launch_floor import chain
rng = np.random.default_rng(0)
for n, outl in ((400, 0.5), (700, 0.8), (1280, 0.3)):
    X, uv, P, inl = synth(rng, n=n, noise=1.0, outliers=outl)
    Xg, ug = torch.from_numpy(X).cuda(), torch.from_numpy(uv).cuda()
    res, R, t, mask = pose.solve_pnp_ransac(Xg, ug, K, iterations=10000)
    rte, rre = pose.get_P_diff(pose.pose_matrix(R, t), P)
    us = chain(lambda: pose.solve_pnp_ransac(Xg, ug, K, iterations=10000), n=10)
    print("n %4d outliers %.1f: success %d inliers %d (true %d)  RTE %.4f m RRE %.4f deg   %.1f us per solve (10000 hypotheses)"
          % (n, outl, int(res[0]), int(res[1]), int(inl.sum()), rte, rre, us))
