#!/usr/bin/env python
"""conv_igemm tiles (FORGE_CONV_TILE=A|B|C|D, read per launch) and the wgrad kernel on the ConvGRU shapes, interleaved rounds in ONE
process. (The setprio / 256x128 / 4-wave variants measured with this tool in round 1 were removed again: DESIGN.md tuning log.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
D, Cc = 32, 128
B = int(os.environ.get("AB_SCENES", "1"))
M = B * D ** 3
x, hbuf, zbuf = torch.randn(M, Cc, device=dev), torch.randn(M, Cc, device=dev), torch.rand(M, Cc, device=dev)
o1, o2 = torch.empty(M, Cc, device=dev), torch.empty(M, Cc, device=dev)
shapes = {"gates": (256, Cc, co.EPI_GRU_GATES), "state": (128, Cc, co.EPI_GRU_OUT), "fconv": (128, 0, co.EPI_AFFINE_ACT)}
ws = {k: torch.randn(27, v[0], Cc + v[1], device=dev) * 0.01 for k, v in shapes.items()}


def run(name):
    Cout, C2, epi = shapes[name]
    bias = torch.zeros(Cout, device=dev)
    co.conv_igemm(x, Cc, Cc, hbuf if C2 else None, C2, C2, ws[name], bias, bias + 1, bias, 0.01, None, hbuf, zbuf, o1,
                  o2 if epi == co.EPI_GRU_GATES else None, (B, D, D, D), (D, D, D), Cout, Cc if epi == co.EPI_GRU_GATES else Cout,
                  co.TAPS_3x3x3, epilogue=epi)


def timeit(name, iters=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run(name)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


# weight-gradient kernel on the same shapes
for name, (Cout, C2, _) in shapes.items():
    dy = torch.randn(M, Cout, device=dev)
    dwp = torch.zeros(27, Cout, Cc + C2, device=dev)
    f = lambda: co.conv_wgrad(dy, x, Cc, hbuf if C2 else None, C2, dwp, (B, D, D, D), (D, D, D), Cout, co.TAPS_3x3x3)
    f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("wgrad %-6s %.3f ms (%.1f TF)" % (name, ms, 2.0 * M * Cout * 27 * (Cc + C2) / ms / 1e9))

variants = os.environ.get("AB_TILES", "A,B,D").split(",")
res = {}
for rnd in range(4):
    for v in variants:
        with co.force_plan(tile=v):
            for name in shapes:
                if rnd == 0:
                    run(name)
                res.setdefault((v, name), []).append(timeit(name))
for (v, name), ts in sorted(res.items()):
    Cout, C2, _ = shapes[name]
    fl = 2.0 * M * Cout * 27 * (Cc + C2)
    best, med = min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]
    print("tile %s %-6s min %.3f ms (%.1f TF)  median %.3f ms (%.1f TF)" % (v, name, best, fl / best / 1e9, med, fl / med / 1e9))
