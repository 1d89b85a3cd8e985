"""Do MFMA-bound and HBM-bound launches that share the chip slow each other down? Stream 1 loops the Winograd gates point-GEMM launch, stream 2 loops
the two transform kernels around it (or a second GEMM loop). Reports the GEMM's rate alone and under each companion."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from forge_amd import convops as co
dev = torch.device("cuda:0")
B, D, C = 1, 32, 128
R, M = B * D * 16 * 16, B * D ** 3
V1, V2 = torch.randn(16, R, C, device=dev), torch.randn(16, R, C, device=dev)
U = torch.randn(16, 3, 256, 2 * C, device=dev) * 0.01
Mm, MmB = torch.empty(16, R, 256, device=dev), torch.empty(16, R, 256, device=dev)
h, z, hr = torch.randn(M, C, device=dev), torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
Vh = torch.empty(16, R, C, device=dev)
bias = torch.zeros(256, device=dev)
MmT = torch.randn(16, R, 256, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
flops = 2.0 * 16 * R * 256 * 3 * 2 * C
gemm = lambda out: co.wino_gemm(V1, C, V2, C, U, out, B, D, 16, 16, 256)


def transforms():
    co.wino_input(h, C, C, B, D, D, D, out=Vh)
    co.wino_output(MmT, bias, None, None, 1.0, None, h, None, z, hr, None, B, D, D, D, 256, C, co.EPI_GRU_GATES)


def run(n, companion):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        for _ in range(n):
            gemm(Mm)
    if companion == "transforms":
        with torch.cuda.stream(s2):
            for _ in range(6 * n):
                transforms()
    elif companion == "gemm":
        with torch.cuda.stream(s2):
            for _ in range(n):
                gemm(MmB)
    e = torch.cuda.Event(); 
    s1.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3


for comp in (None, "transforms", "gemm", None):
    run(5, comp)
    ms, tot = run(40, comp)
    print("GEMM stream with companion %-10s: %.3f ms per GEMM launch = %.1f TF%s" % (comp, ms, flops / ms / 1e9,
          "  (both streams done after %.3f ms per iteration -> %.1f TF aggregate)" % (tot, 2 * flops / tot / 1e9) if comp == "gemm" else ""))
