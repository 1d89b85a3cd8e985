"""Launch sequence for a rocprofv3 --pmc FETCH_SIZE pass: the point GEMM in both forms (16 planes / 8 planes with the row stage in the epilogue) on the
fusion's shapes, 3 launches each, in a fixed order (tools/debug/run_wino_half_fetch.sh reads the per-dispatch counter values back in that order)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
CASES = (("gates [x|h] -> 256", 128, 128, 256), ("state [x|hr] -> 128", 128, 128, 128), ("gates h half 128 -> 256", 128, 0, 256), ("fusion_conv 128 -> 128", 128, 0, 128))
if __name__ == "__main__":
    n, D, Ht, Wt = 1, 32, 16, 16
    R = n * D * Ht * Wt
    g = torch.Generator(device=dev).manual_seed(0)
    for name, C1, C2, Cout in CASES:
        V1 = torch.randn(16, R, C1, device=dev, generator=g)
        V2 = torch.randn(16, R, C2, device=dev, generator=g) if C2 else None
        U = torch.randn(16, 3, Cout, C1 + C2, device=dev, generator=g) * 0.03
        Mm = torch.empty(16, R, Cout, device=dev)
        for half in (False, True):
            for _ in range(3):
                co.wino_gemm(V1, C1, V2, C2, U, Mm, n, D, Ht, Wt, Cout, half=half)
        torch.cuda.synchronize()
