#!/usr/bin/env python
"""Interleaved A/B of forge_conv_igemm environment variants (AB_VARIANTS as in conv_variants.py) on the small-M GEMMs of the ResNet trunk at
one scene (M = 5120 / 20480 rows): median ms over AB_ROUNDS rounds of 20 back-to-back launches each (launch overhead amortised by the
queue), plus the sum over the list weighted by how often each shape occurs in the trunk."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
from _variants import variant  # noqa: E402

dev = torch.device("cuda:0")
# (M, Cout, Cin, taps, occurrences in the trunk)
SHAPES = [(20480, 64, 64, 1, 1), (20480, 64, 64, 9, 3), (20480, 256, 64, 1, 4), (20480, 64, 256, 1, 2), (20480, 128, 256, 1, 1),
          (5120, 128, 128, 9, 4), (5120, 512, 128, 1, 4), (5120, 512, 256, 1, 1), (5120, 128, 512, 1, 3), (5120, 256, 512, 1, 1),
          (5120, 256, 256, 9, 6), (5120, 1024, 256, 1, 6), (5120, 1024, 512, 1, 1), (5120, 256, 1024, 1, 5), (5120, 512, 1024, 1, 1),
          (5120, 512, 512, 9, 3), (5120, 2048, 512, 1, 3), (5120, 2048, 1024, 1, 1), (5120, 512, 2048, 1, 2)]
spec = os.environ.get("AB_VARIANTS", "plan:;nosplit:FORGE_CONV_KSPLIT=1;D:FORGE_CONV_TILE=D;B:FORGE_CONV_TILE=B")
variants = []
for item in spec.split(";"):
    name, _, envs = item.partition(":")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
keys = sorted({k for _, e in variants for k in e})
res = {}
for (M, N, K, T, occ) in SHAPES:
    side = int(round((M / 5) ** 0.5))
    x = torch.randn(M, K, device=dev)
    w = torch.randn(T, N, K, device=dev) * 0.02
    sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    taps = [(0, 0, 0)] if T == 1 else [(0, dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    f = lambda: co.conv_igemm(x, K, K, None, 0, 0, w, None, sc, sh, 0.0, None, None, None, out, None, (5, 1, side, side), (1, side, side), N, N, taps,
                              epilogue=co.EPI_AFFINE_ACT)
    for rnd in range(int(os.environ.get("AB_ROUNDS", "5"))):
        for vname, env in variants:
            with variant(env):
                f()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    f()
                b.record()
                torch.cuda.synchronize()
            res.setdefault((M, N, K, T, occ), {}).setdefault(vname, []).append(a.elapsed_time(b) / 20)
tot = {v: 0.0 for v, _ in variants}
for key, d in res.items():
    M, N, K, T, occ = key
    line = "M=%-6d N=%-5d Cin=%-5d taps=%d x%d plan=%s |" % (M, N, K, T, occ, co.conv_plan(M, N, K, T, co.EPI_AFFINE_ACT, N))
    for v, _ in variants:
        ms = statistics.median(d[v])
        tot[v] += ms * occ
        line += "  %s %.1f us (%.0f TF)" % (v, ms * 1e3, 2.0 * M * N * K * T / ms / 1e9)
    print(line)
print("trunk-weighted sum:", "  ".join("%s %.3f ms" % (v, t) for v, t in tot.items()))
