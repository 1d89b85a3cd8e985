"""Feasibility probe for a full 3-D Winograd F(2,3)^3 (64 points, K = Cin per point, 3.375x fewer multiplies): time the point GEMMs it would
need (64 problems of [R/2 x Cin] x [Cin x N], run as 4 launches of the existing 16-problem kernel with one tap) against the committed
2-D x depth-taps form (16 problems of [R x 3 Cin])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); from _variants import apply_environ  # noqa: E402,E702

dev = torch.device("cuda:0")
B = int(os.environ.get("WINO_SCENES", "1"))
D = int(os.environ.get("WINO_GRID", "32"))


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


for name, C1, C2, N in (("gates", 128, 128, 256), ("state", 128, 128, 128), ("fconv", 128, 0, 128)):
    for tile in ("B", "D"):
        os.environ["FORGE_CONV_TILE"] = tile
        apply_environ()
        # committed form
        R = B * D * (D // 2) * (D // 2)
        V1, V2 = torch.randn(16, R, C1, device=dev), (torch.randn(16, R, C2, device=dev) if C2 else None)
        U = torch.randn(16, 3, N, C1 + C2, device=dev) * 0.02
        Mm = torch.empty(16, R, N, device=dev)
        t2 = timed(lambda: co.wino_gemm(V1, C1, V2, C2, U, Mm, B, D, D // 2, D // 2, N))
        # 3-D form: 64 points x R/2 rows, one tap: 4 launches of 16 points
        R3 = B * (D // 2) ** 3
        W1, W2 = torch.randn(16, R3, C1, device=dev), (torch.randn(16, R3, C2, device=dev) if C2 else None)
        U3 = torch.randn(16, 1, N, C1 + C2, device=dev) * 0.02
        M3 = torch.empty(16, R3, N, device=dev)

        def run3():
            for _ in range(4):
                co.wino_gemm(W1, C1, W2, C2, U3, M3, B, D // 2, D // 2, D // 2, N)
        t3 = timed(run3)
        f2, f3 = 2.0 * 16 * R * N * 3 * (C1 + C2), 2.0 * 64 * R3 * N * (C1 + C2)
        print("scenes %d %-6s tile %s | 2-D x 3 taps: %.3f ms (%.0f TF) | 3-D, 64 points in 4 launches: %.3f ms (%.0f TF) | extra transform traffic ~%.0f MB"
              % (B, name, tile, t2, f2 / t2 / 1e9, t3, f3 / t3 / 1e9, (64 * R3 - 16 * R) * 4 * (C1 + (C2 if name != "fconv" else 0) * 0 + N) / 1e6))
os.environ.pop("FORGE_CONV_TILE", None)
apply_environ()
