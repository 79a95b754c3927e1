import sys, torch, numpy as np
sys.path.insert(0, '.')
from cofii2p_amd.network import CoFiI2P
from cofii2p_amd.preprocess import build_pyramid
from cofii2p_amd.synth import make_frame, subsample_indices
class Opt: img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"
DEV='cuda:0'
model = CoFiI2P(Opt()).to(DEV)
fr = make_frame(11, 4096)
sub = [torch.from_numpy(s).to(DEV) for s in subsample_indices(4096, 5, seed=11)]
pyr = build_pyramid(torch.from_numpy(fr.points).to(DEV), sub)
pyr["feats"] = torch.from_numpy(fr.feats).to(DEV)
img = torch.from_numpy(fr.img)[None].to(DEV)
P = model._pack(torch.device(DEV))
args = (pyr['points'], pyr['neighbors'], pyr['subsampling'], pyr['upsampling'], pyr['feats'], img, 'test', None, None)
e1 = {k: v.clone() for k,v in model._run_device(P, *args).items()}
e2 = {k: v.clone() for k,v in model._run_device(P, *args).items()}
for k in e1: print('eager-eager', k, bool(torch.equal(e1[k], e2[k])))
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    o = model._run_device(P, *args)
g.replay(); torch.cuda.synchronize()
for k in e1:
    a, b = e1[k], o[k]
    print('eager-graph', k, bool(torch.equal(a, b)), float((a.float()-b.float()).abs().max()))
g.replay(); torch.cuda.synchronize()
for k in e1:
    a, b = e1[k], o[k]
    print('eager-graph2', k, bool(torch.equal(a, b)), float((a.float()-b.float()).abs().max()))
