#!/usr/bin/env python
"""rotate_fwd / rotate_bwd (gather) at the grid sizes rotate.py supports (16..128, models/rotate.py:109-123): algorithmic bytes
(read once + write once) / kernel time against the 8 TB/s HBM peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for D, C, n in [(32, 128, 5), (32, 128, 40), (64, 128, 5), (128, 16, 5), (48, 128, 5), (16, 128, 40)]:
    g = torch.Generator().manual_seed(0)
    vox = torch.randn(n, D, D, D, C, generator=g).to(dev).permute(0, 4, 1, 2, 3)
    _, _, poses = syn.orbit_cameras(max(n, 10), 1.5, 15.0)
    xf = torch.zeros(n, 12, device=dev)
    mode = torch.ones(n, dtype=torch.int32, device=dev)
    mode[0] = 0
    # a rigid rotation about y by 20 degrees * i, expressed in normalised grid coordinates
    for i in range(n):
        a = 0.35 * i
        R = torch.tensor([[torch.cos(torch.tensor(a)), 0, torch.sin(torch.tensor(a))], [0, 1, 0], [-torch.sin(torch.tensor(a)), 0, torch.cos(torch.tensor(a))]])
        xf[i] = torch.cat([R, torch.tensor([[0.01], [0.0], [-0.02]])], dim=1).reshape(12)
    out = ops.rotate_warp(vox, xf, mode)
    ms_f = timeit(lambda: ops.rotate_warp(vox, xf, mode))
    v2 = vox.detach().clone().requires_grad_(True)
    o2 = ops.rotate_warp(v2, xf, mode)
    gy = torch.randn_like(o2)
    ms_b = timeit(lambda: torch.autograd.grad(o2, v2, gy, retain_graph=True))
    by = 2.0 * n * C * D ** 3 * 4
    print("D=%3d C=%3d n=%2d (%.0f MB r+w): fwd %.3f ms = %.2f TB/s (%.2f of 8 TB/s) | bwd(gather) %.3f ms = %.2f TB/s" % (
        D, C, n, by / 1e6, ms_f, by / ms_f / 1e9, by / ms_f / 1e9 / 8.0, ms_b, by / ms_b / 1e9))
