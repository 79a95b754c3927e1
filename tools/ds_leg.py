"""GPU tool: the loader feeding stack-mode submissions (bench.loader_stack_pipeline) at several draw-worker counts / slot depths.
    python tools/ds_leg.py  [DS_WORKERS=4,8,16  DS_KS=4,8]"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))


def main():
    import torch, bench
    from cofii2p_amd import ops
    from cofii2p_amd.network import CoFiI2P
    dev = torch.device("cuda", 0)
    ops.GEMM_MODE = "bf16x6"
    CoFiI2P.MAX_STABLE_GRAPHS = 512
    model = CoFiI2P(bench.Opt()).to(dev); bench._model_ref.append(model); model.enable_graphs(True)
    opt_ds, raw, raw_d, img_d, rK, P_Tr = bench.dataside_inputs(dev, 20480)
    st = bench.make_streams(dev, 4)
    sb = 160
    for W in [int(x) for x in os.environ.get("DS_WORKERS", "4,8,16").split(",")]:
        for KS in [int(x) for x in os.environ.get("DS_KS", "4").split(",")]:
            for upk in (None, 1):
                r, h = bench.loader_stack_pipeline(model, dev, opt_ds, st, 16, 32, upk, raw_d, img_d, rK, P_Tr, sb, workers=W, KS=KS)
                sb += 20
                print("workers", W, "KS", KS, "upk", upk, round(r, 1), "f/s", {k: round(1e3 * v, 3) for k, v in h.items()}, flush=True)


if __name__ == "__main__":
    main()
