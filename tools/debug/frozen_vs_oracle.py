import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import forge_oracle as fo
from forge_amd import synthetic as syn
from forge_amd.model import FORGE
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE(cfg)
w = syn.seeded_state_dict(model.state_dict(), 0)
model.load_state_dict(w)
model = model.to(dev).eval()
g = torch.Generator().manual_seed(17)
x0 = (torch.randn(1, 3, 128, 16, 16, 16, generator=g) * 0.5)
rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)

def loss_of(fused, feat, dens, which):
    wf = torch.linspace(-1, 1, feat.numel(), device=feat.device, dtype=feat.dtype).reshape(feat.shape)
    wd = torch.linspace(1, -1, dens.numel(), device=dens.device, dtype=dens.dtype).reshape(dens.shape)
    terms = {"feat": (feat * wf).sum(), "dens": (dens * wd).sum(), "fused": fused.square().sum() * 1e-3}
    return terms[which] if which != "all" else sum(terms.values())

for which in ("fused", "feat", "dens", "all"):
    res = {}
    for frozen in (True, False):
        for p in model.parameters():
            p.requires_grad_(not frozen)
        x = x0.clone().to(dev).requires_grad_(True)
        fused = model.encoder_3d.fuse(x)
        feat, dens = model.encoder_3d.heads(fused)
        loss_of(fused, feat, dens, which).backward()
        res[frozen] = x.grad.cpu().double()
    wd = {k: v.double() for k, v in w.items()}
    xo = x0.clone().double().requires_grad_(True)
    fo_f = fo.fuse(xo, wd)
    loss_of(fo_f, fo.render_features_head(fo_f, wd), fo.density_head(fo_f, wd), which).backward()
    print("%-6s frozen-vs-oracle %.3e   autograd-vs-oracle %.3e   frozen-vs-autograd %.3e" % (which, rel(res[True], xo.grad), rel(res[False], xo.grad), rel(res[True], res[False])))
    d = (res[True] - xo.grad).abs()
    i = d.argmax().item()
    idx = torch.unravel_index(torch.tensor(i), d.shape)
    print("       worst frozen idx", [int(v) for v in idx], "per-view rel err", [rel(res[True][:, t], xo.grad[:, t]) for t in range(3)])
