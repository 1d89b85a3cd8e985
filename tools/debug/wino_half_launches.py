"""Per-shape time of the point GEMM + inverse transform pair in both forms (16 planes / 8 planes with the row stage in the GEMM epilogue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, (n, D, H, W, C1, C2, Cout) in (("gates [x|h] -> 256", (1, 32, 32, 32, 128, 128, 256)), ("state [x|hr] -> 128", (1, 32, 32, 32, 128, 128, 128)),
                                         ("fusion_conv 128 -> 128", (1, 32, 32, 32, 128, 0, 128)), ("conv1 5 views 64 -> 128", (5, 32, 32, 32, 64, 0, 128)),
                                         ("gates, 8 scenes", (8, 32, 32, 32, 128, 128, 256)), ("state, 8 scenes", (8, 32, 32, 32, 128, 128, 128))):
    g = torch.Generator(device=dev).manual_seed(1)
    R, M = n * D * (H // 2) * (W // 2), n * D * H * W
    V1 = torch.randn(16, R, C1, device=dev, generator=g)
    V2 = torch.randn(16, R, C2, device=dev, generator=g) if C2 else None
    U = torch.randn(16, 3, Cout, C1 + C2, device=dev, generator=g) * 0.03
    Mm = torch.empty(16, R, Cout, device=dev)
    out = torch.empty(M, Cout, device=dev)
    bias = torch.zeros(Cout, device=dev)
    res = {}
    for half in (False, True):
        tg = timed(lambda: co.wino_gemm(V1, C1, V2, C2, U, Mm, n, D, H // 2, W // 2, Cout, half=half))
        to = timed(lambda: co.wino_output(Mm, bias, None, None, 1.0, None, None, None, out, None, None, n, D, H, W, Cout, Cout, co.EPI_BIAS, half=half))
        res[half] = (tg, to)
    print("%-26s 16 planes: GEMM %7.1f us + inverse %6.1f us = %7.1f | 8 planes: GEMM %7.1f us + inverse %6.1f us = %7.1f  (x%.3f)" %
          (name, res[False][0], res[False][1], sum(res[False]), res[True][0], res[True][1], sum(res[True]), sum(res[False]) / sum(res[True])), flush=True)
