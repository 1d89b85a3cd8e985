#!/usr/bin/env python
"""Time forge_render_bwd (10 views of 128^2 rays x 64 samples on a 64^3 x 16 volume, and 4 views on 128^3) on a dense synthetic
volume (Gaussian blob density, every sample inside the blob scatters) and on a nearly empty one (random-init heads: the
zero-gradient skips remove most of the scatter)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
for D, V in [(64, 10), (128, 4), (32, 10)]:
    C, Hr, S = 16, 128, 64
    feat, dens = syn.blob_volumes(1, D, C, seed=3)
    _, extr, _ = syn.orbit_cameras(10, 1.5, 15.0)
    E = extr[:V]
    K = syn.intrinsics(256) / 2.0
    cam = torch.cat([E[:, :3, :3].reshape(V, 9), E[:, :3, 3], K[0, 0].expand(V, 1), K[1, 1].expand(V, 1), K[0, 2].expand(V, 1),
                     K[1, 2].expand(V, 1)], dim=1).contiguous().to(dev)
    v2v = torch.zeros(V, dtype=torch.int32, device=dev)
    h = [0.5 * (D - 1) / D] * 3
    for name, dn in (("dense blob", dens), ("sparse (1% of the blob)", dens * (torch.rand_like(dens) < 0.01))):
        for with_cam in (False, True):
            f = feat.to(dev).requires_grad_(True)
            d = dn.to(dev).requires_grad_(True)
            c = cam.clone().requires_grad_(with_cam)
            of, oo = ops.render_rays(f, d, c, v2v, Hr, Hr, S, 0.5, 2.0, h, False)
            gf, go = torch.randn_like(of), torch.randn_like(oo)
            ts = []
            for it in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f.grad = d.grad = None
                torch.cuda.synchronize()
                a.record()
                torch.autograd.backward([of, oo], [gf, go], retain_graph=True)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            print("D=%3d V=%2d %-24s cam-grad=%d: backward %.3f ms (incl. 2 zero-fills)" % (D, V, name, with_cam, min(ts[1:])))
