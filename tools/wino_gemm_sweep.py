#!/usr/bin/env python
"""forge_wino_gemm (16 batched 3-tap point GEMMs, one conv_igemm_kernel launch) at the fusion's shapes under every tile override
(FORGE_CONV_TILE): ms and MFMA TF per tile, and what the plan model picks. WINO_SCENES, WINO_GRID."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("WINO_SCENES", "1"))
D = int(os.environ.get("WINO_GRID", "32"))
shapes = [("gates  C=128+128 N=256", B, D, 128, 128, 256), ("state  C=128+128 N=128", B, D, 128, 128, 128), ("fconv  C=128     N=128", B, D, 128, 0, 128),
          ("xhalf  C=128     N=256 (5 views)", 5 * B, D, 128, 0, 256), ("conv1  C=64      N=128 (5 views)", 5 * B, D, 64, 0, 128)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, n, D_, C1, C2, N in shapes:
    R = n * D_ * (D_ // 2) * (D_ // 2)
    V1 = torch.randn(16, R, C1, device=dev)
    V2 = torch.randn(16, R, C2, device=dev) if C2 else None
    U = torch.randn(16, 3, N, C1 + C2, device=dev) * 0.02
    Mm = torch.empty(16, R, N, device=dev)
    flops = 2.0 * 16 * R * N * 3 * (C1 + C2)
    plan = co.wino_gemm_tile(R, N, C1 + C2)
    line = []
    for tile in os.environ.get("SWEEP_TILES", "A,B,C,D").split(","):
        with co.force_plan(tile=tile):
            ms = timed(lambda: co.wino_gemm(V1, C1, V2, C2, U, Mm, n, D_, D_ // 2, D_ // 2, N))
        line.append("%s %.3f ms %5.1f TF" % (tile, ms, flops / ms / 1e9))
    print("scenes %d  %-36s plan %s | %s" % (B, name, plan, " | ".join(line)))
