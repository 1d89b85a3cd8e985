"""Ceiling experiments on the MFMA K-loop: the Winograd gates point-GEMM launch (16 x [8192 x 768] x [768 x 256]) and the direct gates launch
(M = 32768, N = 256, K = 6912) on a variant library (GEMM_LIB = name built by tools/debug/build_variant_lib.sh; results of the variants are
numerically meaningless, only the time counts). Prints ms and TF per tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = os.environ.get("GEMM_LIB", "")
if name:
    os.environ["FORGE_AMD_LIB"] = os.path.join(ROOT, "tools", "debug", "libforge_hip_%s.so" % name)
sys.path.insert(0, ROOT)
import torch
from forge_amd import convops as co
dev = torch.device("cuda:0")


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


B, D, C = 1, 32, 128
R = B * D * (D // 2) * (D // 2)
V1, V2 = torch.randn(16, R, C, device=dev), torch.randn(16, R, C, device=dev)
res = []
for N in (256, 128):
    U = torch.randn(16, 3, N, 2 * C, device=dev) * 0.01
    Mm = torch.empty(16, R, N, device=dev)
    fl = 2.0 * 16 * R * N * 3 * 2 * C
    for tile in os.environ.get("GEMM_TILES", "DBA"):
        with co.force_plan(tile=tile):
            ms = timed(lambda: co.wino_gemm(V1, C, V2, C, U, Mm, B, D, D // 2, D // 2, N))
        res.append("wino N=%d tile %s %.3f ms %.1f TF" % (N, tile, ms, fl / ms / 1e9))
M = B * D ** 3
x, h = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
o1, o2 = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
w = torch.randn(27, 256, 2 * C, device=dev) * 0.01
bias = torch.zeros(256, device=dev)
for tile in os.environ.get("GEMM_TILES", "DBA"):
    with co.force_plan(tile=tile, ksplit=1):
        ms = timed(lambda: co.conv_igemm(x, C, C, h, C, C, w, bias, None, None, 1.0, None, h, None, o1, o2, (B, D, D, D), (D, D, D), 256, C, co.TAPS_3x3x3,
                                         epilogue=co.EPI_GRU_GATES), reps=4)
    res.append("direct gates tile %s %.3f ms %.1f TF" % (tile, ms, 2.0 * M * 256 * 27 * 2 * C / ms / 1e9))
print("[%s] " % (name or "product") + " | ".join(res))
