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
    for KS in (4, 8):
        for upk in (1,):
            r, h = bench.loader_stack_pipeline(model, dev, opt_ds, st, 16, 32, upk, raw_d, img_d, rK, P_Tr, 160 if upk is None else 180, KS=KS)
            print("stack KS", KS, "upk", upk, round(r, 1), {k: round(1e3 * v, 3) for k, v in h.items()})
    for upk in ():
        r = bench.loader_pipeline(model, dev, opt_ds, st, 2, 200, upk, raw_d, img_d, rK, P_Tr, 60 if upk is None else 80)
        print("batch1 upk", upk, round(r[0], 1))


if __name__ == "__main__":
    main()
