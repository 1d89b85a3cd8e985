#!/usr/bin/env python
"""forge_render_fwd at the big-volume shapes: D_r in {64, 128}, V in {5 (bench), 28 (360-degree NVS, kubric_eval.py:166-232)} views of ONE
volume, 128^2 rays x 64 samples. Prints ms per launch and G taps/s; RENDER_PROBE_ITERS launches each (PMC target: tools/pmc_render.sh).
(The XCD-contiguous tile order and the wave-per-ray variant measured with this tool in round 2 - profiles/r02_render_ab.txt - are no longer in the library.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import _lib, nvs, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
lib, st = _lib.lib(), _lib.current_stream()
iters = int(os.environ.get("RENDER_PROBE_ITERS", "10"))
# image rows: 128 = the product shape (16 tile rows -> the XCD band order of forge_render_fwd); 120 = 15 tile rows, which the library launches in plain
# launch order (ny % 8 != 0): the same kernel and volume WITHOUT the band placement, for the traffic comparison of tools/pmc_render.sh
Hr = int(os.environ.get("RENDER_PROBE_HR", "128"))
cases = [tuple(int(v) for v in c.split("x")) for c in os.environ.get("RENDER_PROBE_CASES", "64x5,64x28,128x5,128x28").split(",")]
for Dr, V in cases:
    feat, dens = syn.blob_volumes(1, Dr, 16, seed=0)
    feat = feat.to(dev).permute(0, 2, 3, 4, 1).contiguous()
    dens = dens.to(dev).contiguous()
    _, extr, _ = syn.orbit_cameras(10, 1.5, 10.0)
    E = torch.stack([extr[i % 10] for i in range(V)])
    if V > 10:                                                       # a ring of V cameras about the object centre
        import math
        Tc = torch.eye(4); Tc[2, 3] = 1.5
        E = torch.stack([torch.inverse(torch.inverse(Tc) @ torch.inverse(syn._rot_y(2 * math.pi * i / V) @ syn._rot_x(0.3)) @ Tc @ torch.inverse(Tc)) for i in range(V)])
        E = torch.stack([Tc @ syn._rot_y(2 * math.pi * i / V) @ syn._rot_x(0.3) @ torch.inverse(Tc) @ Tc for i in range(V)])
    K = syn.intrinsics(256) / 2.0
    cam = torch.cat([E[:, :3, :3].reshape(V, 9), E[:, :3, 3], K[0, 0].expand(V, 1), K[1, 1].expand(V, 1), K[0, 2].expand(V, 1), K[1, 2].expand(V, 1)],
                    dim=1).contiguous().to(dev)
    v2v = torch.zeros(V, dtype=torch.int32, device=dev)
    of, oo = torch.empty(V, Hr, 128, 16, device=dev), torch.empty(V, Hr, 128, device=dev)
    h = 0.5 * (Dr - 1) / Dr
    f = lambda: _lib.check(lib.forge_render_fwd(_lib.ptr(feat), _lib.ptr(dens), _lib.ptr(cam), _lib.ptr(v2v), _lib.ptr(of), _lib.ptr(oo), None,
                                                V, 1, 16, Dr, Dr, Dr, Hr, 128, 64, 0.5, 2.0, h, h, h, st), "render")
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    hit = (oo > 0).float().mean().item()
    print("render D_r=%d V=%d rows=%d: %.4f ms/launch  %.1f us/view  %.2f G taps/s  (opacity>0 on %.0f%% of the rays; workgroup order: %s)"
          % (Dr, V, Hr, ms, ms * 1e3 / V, V * Hr * 128 * 64 * 17 * 8 / ms / 1e6, 100 * hit, "XCD row bands" if (Hr // 8) % 8 == 0 else "launch order"))
