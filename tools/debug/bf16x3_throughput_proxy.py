#!/usr/bin/env python
"""VERDICT r5 item 3, priced by MEASUREMENT before a kernel is written: what would a 3 x bf16 split of the fusion's Winograd point GEMMs buy?

The ConvGRU gates launch is 16 x ([8192 x 768] x [768 x 256]) in fp32 (51.5 GF executed, forge wino_gemm). The split form executes SIX bf16 products
per fp32 product (a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0, fp32 accumulate). Proxy for what the 2.5 PF pipe delivers at exactly this problem
geometry: the vendor-tuned bf16 GEMM (torch.bmm -> hipBLASLt / rocBLAS) on the six products written as ONE GEMM with the planes concatenated along
K: 16 x ([8192 x 4608] x [4608 x 256]); and the 3-product truncation (K = 2304). A hand-written kernel reads 3 planes for 6 products (less LDS / HBM
traffic than the concatenated form) but would have to beat a library kernel's schedule to be faster than this proxy; the guide's own 256^2
template runs 1320-1340 TF on random operands at 4096^3.

Also timed: what the split costs outside the GEMM (three bf16 planes of the transformed operand written instead of one fp32 plane)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


B, D, C = 1, 32, 128
R = B * D * (D // 2) * (D // 2)
print("one scene: R = %d rows per Winograd point, 16 points" % R, flush=True)
for name, Cout, C2 in (("gates N=256 K=768", 256, C), ("state N=128 K=768", 128, C), ("fusion_conv N=128 K=384", 128, 0)):
    K = 3 * (C + C2)
    V1, V2 = torch.randn(16, R, C, device=dev), torch.randn(16, R, C, device=dev)
    U = torch.randn(16, 3, Cout, C + C2, device=dev) * 0.02
    Mm = torch.empty(16, R, Cout, device=dev)
    ms32 = timed(lambda: co.wino_gemm(V1, C, V2 if C2 else None, C2, U, Mm, B, D, D // 2, D // 2, Cout))
    gf = 2.0 * 16 * R * Cout * K / 1e9
    print("%-26s exact fp32 (forge wino_gemm): %.3f ms  %.1f TF executed" % (name, ms32, gf / ms32))
    for terms in (6, 3):
        # ONE 2-D library GEMM over all 16 points' rows ([16 R x terms K] x [terms K x N], the weight shared instead of per point: the same MFMA work and
        # operand traffic per row, and the grid the real launch has - 16 separate GEMMs of 8192 rows each fill half the chip; torch.bmm(out=) faulted here)
        A = torch.randn(16 * R, terms * K, device=dev).to(torch.bfloat16)
        Bm = (torch.randn(terms * K, Cout, device=dev) * 0.02).to(torch.bfloat16)
        Bt = Bm.t().contiguous()                                      # [N, K]: K-contiguous for both operands, what an MFMA kernel stages
        ms = timed(lambda: torch.matmul(A, Bm), iters=10, warm=3)
        ms_t = timed(lambda: torch.matmul(A, Bt.t()), iters=10, warm=3)
        best = min(ms, ms_t)
        print("    %d bf16 products as one library GEMM (K = %d): %.3f ms (B as [K,N]) / %.3f ms (B as [N,K])  -> %.0f TF bf16;  vs exact fp32: %.2fx"
              % (terms, terms * K, ms, ms_t, terms * gf / best, ms32 / best), flush=True)
        del A, Bm, Bt
    # the split's cost outside the GEMM: three bf16 planes of the transformed operand instead of one fp32 plane (written by the input transform, here as
    # a stand-alone pass: read fp32, write 3 x bf16)
    X = torch.randn(16, R, C + C2, device=dev)

    def split():
        a0 = X.to(torch.bfloat16)
        r1 = X - a0.float()
        a1 = r1.to(torch.bfloat16)
        return a0, a1, (r1 - a1.float()).to(torch.bfloat16)
    ms_s = timed(split)
    byts = X.numel() * (4 + 6)
    print("    split of the transformed operand as separate torch passes: %.3f ms; as a fused epilogue it is %.1f MB of traffic = %.3f ms at 4 TB/s"
          % (ms_s, byts / 1e6, byts / 4e12 * 1e3))
print("kill criterion (VERDICT r5 item 3): < 1.15x on the one-scene step. The point GEMMs are ~4.0 ms of the 6.5 ms step; a GEMM speed-up s gives a step "
      "speed-up of 1 / (1 - 0.61 (1 - 1/s)) at best (the transforms and the split's extra bytes not yet charged)")
