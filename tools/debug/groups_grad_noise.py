"""Shared-input-halves vs separate fusions under autograd (train-mode BN): input-gradient agreement per seed, direct and Winograd kernels.
Jumps to ~1e-4 are single LeakyReLU arguments within rounding of zero taking the other slope in one of the two paths."""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forge_amd import synthetic as syn
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); from _variants import apply_environ  # noqa: E402,E702
from forge_amd.fusion import ConvGRU_3D
dev = torch.device("cuda:0")
for seed, mode in [(sd, m) for sd in (11, 12, 13, 14, 15) for m in ("0", "1")]:
    os.environ["FORGE_WINOGRAD"] = mode
    apply_environ()
    torch.manual_seed(seed)
    a = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=128, hidden_size=128).to(dev).train()
    bmod = copy.deepcopy(a)
    x = (torch.randn(1, 5, 128, 16, 16, 16) * 0.5).to(dev)
    groups = [[0, 1, 2], [3, 4], [0, 1, 2, 3, 4]]
    ws = [torch.randn(1, 128, 16, 16, 16, device=dev) for _ in groups]
    xa = x.clone().requires_grad_(True)
    outs_a = a.fuse_groups_autograd_hip(xa, groups)
    sum((o * w).sum() for o, w in zip(outs_a, ws)).backward()
    xb = x.clone().requires_grad_(True)
    outs_b = [bmod.fuse_autograd_hip(xb[:, g]) for g in groups]
    sum((o * w).sum() for o, w in zip(outs_b, ws)).backward()
    d = (xa.grad - xb.grad)
    print("seed", seed, "mode", mode, "rel L2 %.3e  max rel %.3e" % ((d.norm() / xb.grad.norm()).item(), (d.abs().max() / xb.grad.abs().max()).item()),
          "out rel", [((oa - ob).norm() / ob.norm()).item() for oa, ob in zip(outs_a, outs_b)])
    if mode == "0":
        keep = (xa.grad.clone(), xb.grad.clone())
    else:
        print("winograd-vs-direct shared: %.3e   separate: %.3e" % (((xa.grad - keep[0]).norm() / keep[0].norm()).item(), ((xb.grad - keep[1]).norm() / keep[1].norm()).item()))
