#!/usr/bin/env python
"""Sweep (tile, split-K) of forge_conv_igemm over the small-M shapes of the step (ResNet-50 trunk at 1 and 4 scenes, heads) and
compare the measured best with the choice of the launch-plan model (csrc/conv_igemm.hip: plan_conv). The overrides
FORGE_CONV_TILE / FORGE_CONV_KSPLIT are read per launch, so everything runs interleaved in ONE process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # (M, Cout, Cin, taps)  2-D shapes are (n, 1, H, W) grids
    (5120, 2048, 512, 1), (5120, 512, 2048, 1), (5120, 512, 512, 9), (5120, 2048, 1024, 1), (5120, 512, 1024, 1),
    (5120, 1024, 256, 1), (5120, 256, 1024, 1), (5120, 256, 256, 9), (5120, 1024, 512, 1), (5120, 256, 512, 1),
    (5120, 512, 128, 1), (5120, 128, 512, 1), (5120, 128, 128, 9), (5120, 512, 256, 1),
    (20480, 256, 64, 1), (20480, 64, 256, 1), (20480, 64, 64, 9), (20480, 64, 64, 1), (20480, 128, 256, 1),
    (81920, 64, 160, 1),
    (32768, 64, 128, 8),            # ConvTranspose3d phase GEMM (both heads share it: 64 = 2 x 32 channels)
    (32768, 128, 128, 27), (32768, 128, 256, 27), (32768, 256, 256, 27), (163840, 128, 64, 27),
    (786432, 32, 32, 27), (98304, 32, 128, 8),   # training-path heads
]
if os.environ.get("SWEEP_SCENES", "1") != "1":
    k = int(os.environ["SWEEP_SCENES"])
    SHAPES = [(M * k, N, C, T) for (M, N, C, T) in SHAPES if M * k * max(N, C) * 4 < (1 << 31)]
TILES = os.environ.get("SWEEP_TILES", "ABCDE")
SPLITS = [1, 2, 3, 4, 6, 8]


def setup(M, N, C, T):
    if T == 27:
        side = 64 if M % (64 ** 3) == 0 and N <= 32 else 32
        n = M // side ** 3
        assert n * side ** 3 == M, (M, n, side)
        grid = (n, side, side, side)
        taps = co.TAPS_3x3x3
    elif T == 8:
        n = M // 32768
        assert n * 32768 == M
        grid = (n, 32, 32, 32)
        taps = [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]
    else:
        n = 5 * max(1, M // (5 * 128 * 128)) if M % 5 == 0 else 1
        hw = M // n
        side = round(hw ** 0.5)
        assert n * side * side == M, (M, n, side)
        grid = (n, 1, side, side)
        taps = [(0, 0, 0)] if T == 1 else [(0, dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    x = torch.randn(M, C, device=dev)
    w = torch.randn(T, N, C, device=dev) * 0.02
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)

    def run():
        co.conv_igemm(x, C, C, None, 0, 0, w, b, b + 1, b, 0.0, None, None, None, out, None, grid, grid[1:], N, N, taps,
                      epilogue=co.EPI_AFFINE_ACT)
    return run


def timeit(run, iters=12):
    run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("%8s %5s %5s %4s | %-8s %8s %7s | %-8s %8s %7s | %5s | per-tile best (ksplit: us)" % ("M", "N", "Cin", "taps", "model", "us", "TF", "best", "us", "TF", "loss"))
tot_model = tot_best = 0.0
ALL = []
for (M, N, C, T) in SHAPES:
    run = setup(M, N, C, T)
    fl = 2.0 * M * N * C * T
    model = co.conv_plan(M, N, C, T, co.EPI_AFFINE_ACT, N)
    res = {}
    for rnd in range(2):
        t = timeit(run)
        res[("model",)] = min(res.get(("model",), 1e9), t)
        for tile in TILES:
            if N <= 32 and tile in "ABGI":
                continue
            for k in SPLITS:
                with co.force_plan(tile, k):
                    if co.conv_plan(M, N, C, T, co.EPI_AFFINE_ACT, N) != (tile, k):
                        continue                                   # combination not admissible (K too short / workspace)
                    t = timeit(run)
                res[(tile, k)] = min(res.get((tile, k), 1e9), t)
    tm = res.pop(("model",))
    ALL.append({"shape": [M, N, C, T], "model": list(model), "model_us": tm * 1e3, "us": {"%s%d" % k: v * 1e3 for k, v in res.items()}})
    (bt, bk), tb = min(res.items(), key=lambda kv: kv[1])
    per_tile = []
    for tile in TILES:
        c = {k: v for (t_, k), v in res.items() if t_ == tile}
        if c:
            kb = min(c, key=c.get)
            per_tile.append("%s %d:%.0f" % (tile, kb, c[kb] * 1e3))
    tot_model += tm
    tot_best += tb
    print("%8d %5d %5d %4d | %-8s %8.1f %7.1f | %-8s %8.1f %7.1f | %4.0f%% | %s" % (
        M, N, C, T, "%s/%d" % model, tm * 1e3, fl / tm / 1e9, "%s/%d" % (bt, bk), tb * 1e3, fl / tb / 1e9, 100 * (tm / tb - 1), "  ".join(per_tile)))
import json  # noqa: E402
json.dump(ALL, open(os.environ.get("SWEEP_JSON", "gpurun_out/plan_sweep.json"), "w"))
print("sum over shapes: model %.1f us, best %.1f us" % (tot_model * 1e3, tot_best * 1e3))
