import sys, torch, numpy as np
sys.path.insert(0, '.')
from cofii2p_amd.network import CoFiI2P
from cofii2p_amd import image, kpfpn
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
def run():
    taps = {}
    pc = kpfpn.run_fpn(P, pyr['points'], pyr['neighbors'], pyr['subsampling'], pyr['upsampling'], pyr['feats'], taps=taps)
    im = image.resnet34(P, img)
    taps.update({'img%d'%i: t for i,t in enumerate(im)})
    return {k: v.clone() for k,v in taps.items()}
a = run(); b = run(); c = run()
for k in a:
    print(k, bool(torch.equal(a[k], b[k])), bool(torch.equal(a[k], c[k])), float((a[k]-b[k]).abs().max()))
