"""Is the 1.2 % relative-L2 gap between the HIP and the fp32-oracle input-feature gradient of the 64^3 training step noise or a bug?
Third opinion: the same step through the oracle in float64. Prints the three pairwise relative L2 errors."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import forge_oracle as fo
from forge_amd import geo_utils, synthetic as syn
from forge_amd.model import FORGE
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE(cfg)
w = syn.seeded_state_dict(model.state_dict(), 0)
model.load_state_dict(w)
model = model.to(dev).train()
t = 2
g = torch.Generator().manual_seed(13)
feats = torch.randn(1, t, 128, 64, 64, 64, generator=g) * 0.5
jit = (torch.rand(10, 2, generator=g) - 0.5) * 0.3
poses, extr, _ = syn.orbit_cameras(10, 1.5, 12.0, jit)
P, E = poses[None, :t].contiguous(), extr[None, [0, 3]].contiguous()
K = syn.intrinsics(256)[None, None].repeat(1, 2, 1, 1)
g2 = torch.Generator().manual_seed(3)
tgt_i, tgt_m = torch.rand(2, 3, 256, 256, generator=g2), torch.rand(2, 1, 256, 256, generator=g2)
fd = feats.to(dev).requires_grad_(True)
imgs, masks, _ = model.reconstruct(fd, P.to(dev), geo_utils.camera_dict(E.to(dev), K.to(dev)))
(5.0 * torch.nn.functional.mse_loss(imgs, tgt_i.to(dev)) + torch.nn.functional.mse_loss(masks, tgt_m.to(dev))).backward()
hip = fd.grad.cpu().double()
res = {}
for dt in (torch.float32, torch.float64):
    t0 = time.time()
    wd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in w.items()}
    fr = feats.clone().to(dt).requires_grad_(True)
    oi, om = fo.reconstruct_from_features(fr, P.to(dt), E.to(dt), K.to(dt), wd, cfg, training=True, order_by_distance=True)
    (5.0 * torch.nn.functional.mse_loss(oi, tgt_i.to(dt)) + torch.nn.functional.mse_loss(om, tgt_m.to(dt))).backward()
    res[dt] = fr.grad.double()
    print(dt, "oracle pass %.0f s" % (time.time() - t0), flush=True)
rl2 = lambda a, b: ((a - b).norm() / b.norm()).item()
print("rel L2  hip vs f32-oracle %.3e | hip vs f64-oracle %.3e | f32-oracle vs f64-oracle %.3e" % (rl2(hip, res[torch.float32]), rl2(hip, res[torch.float64]), rl2(res[torch.float32], res[torch.float64])))
