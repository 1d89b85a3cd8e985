#!/usr/bin/env python
"""Interleaved A/B of forge_conv_igemm launch variants selected by variant switches (tools/_variants.py: explicit plan arguments, no environment reads in the library) on the
ConvGRU shapes: AB_VARIANTS="name:ENV=VAL,ENV=VAL;name2:..." (default: the plan's choice vs each forced tile; the round-2 experiments
(priority, deeper prefetch, dual accumulators, chunk-outer 128x64 tile) were run through it with switches that no longer exist). Prints ms and TFLOP/s per
(variant, shape), median of AB_ROUNDS interleaved rounds.   AB_SCENES=1|4|8 sets M = scenes * 32^3."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
from _variants import variant  # noqa: E402

dev = torch.device("cuda:0")
D, Cc = 32, 128
B = int(os.environ.get("AB_SCENES", "1"))
M = B * D ** 3
x, hbuf, zbuf = torch.randn(M, Cc, device=dev), torch.randn(M, Cc, device=dev), torch.rand(M, Cc, device=dev)
o1, o2 = torch.empty(M, Cc, device=dev), torch.empty(M, Cc, device=dev)
shapes = {"gates": (256, Cc, co.EPI_GRU_GATES), "state": (128, Cc, co.EPI_GRU_OUT), "fconv": (128, 0, co.EPI_AFFINE_ACT)}
ws = {k: torch.randn(27, v[0], Cc + v[1], device=dev) * 0.01 for k, v in shapes.items()}
spec = os.environ.get("AB_VARIANTS", "plan:;A:FORGE_CONV_TILE=A;B:FORGE_CONV_TILE=B;C:FORGE_CONV_TILE=C;D:FORGE_CONV_TILE=D")
variants = []
for item in spec.split(";"):
    name, _, envs = item.partition(":")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
keys = sorted({k for _, e in variants for k in e})


def run(name):
    Cout, C2, epi = shapes[name]
    bias = torch.zeros(Cout, device=dev)
    co.conv_igemm(x, Cc, Cc, hbuf if C2 else None, C2, C2, ws[name], bias, bias + 1, bias, 0.01, None, hbuf, zbuf, o1,
                  o2 if epi == co.EPI_GRU_GATES else None, (B, D, D, D), (D, D, D), Cout, Cc if epi == co.EPI_GRU_GATES else Cout,
                  co.TAPS_3x3x3, epilogue=epi)


def timeit(name, iters=8):
    run(name)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run(name)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


res = {}
for rnd in range(int(os.environ.get("AB_ROUNDS", "5"))):
    for vname, env in variants:
        with variant(env):
            for s in shapes:
                res.setdefault((vname, s), []).append(timeit(s))
for (vname, s), v in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    Cout, C2, _ = shapes[s]
    ms = statistics.median(v)
    print("scenes %d %-6s %-12s %.4f ms  %.1f TF  (min %.4f)" % (B, s, vname, ms, 2.0 * M * Cout * 27 * (Cc + C2) / ms / 1e9, min(v)))
