#!/usr/bin/env python
"""VERDICT r4 item 9, the numerical half of the question: how far is a 3-term bf16 split of the fp32 operands (a = a0 + a1 + a2, each bf16; the six
cross products a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 accumulated in fp32 - what the 2.5 PF bf16 MFMA pipe would execute) from a float64 product,
next to the exact-fp32 kernel of this repo, on the fusion's Winograd point-GEMM shape ([8192 x 768] x [768 x 256], Winograd-domain operands of
realistic scale)? Emulation: the split operands are bf16 VALUES held in fp32 tensors, the six partial products run as fp32 GEMMs (every bf16 x bf16
product is exact in fp32, the accumulation is fp32 as in the MFMA) - the hardware's own accumulation order would differ in the last bits only.
Also the 4-term (a0b0 + a0b1 + a1b0 + a1b1) and 3-term (a0b0 + a0b1 + a1b0) truncations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.backends.cuda.matmul.allow_tf32 = False


def split3(x):
    a0 = x.to(torch.bfloat16).float()
    a1 = (x - a0).to(torch.bfloat16).float()
    a2 = (x - a0 - a1).to(torch.bfloat16).float()
    return a0, a1, a2


B, D, C, N = 1, 32, 128, 256
R = B * D * (D // 2) * (D // 2)
# Winograd-domain operands: V = B^T x B of activations ~ N(0, 0.5), U = G w G^T of weights ~ N(0, 0.02)
V1, V2 = torch.randn(16, R, C, device=dev) * 1.0, torch.randn(16, R, C, device=dev) * 1.0
U = torch.randn(16, 3, N, 2 * C, device=dev) * 0.02
Mm = torch.empty(16, R, N, device=dev)
co.wino_gemm(V1, C, V2, C, U, Mm, B, D, D // 2, D // 2, N)
torch.cuda.synchronize()
# one point, the centre depth tap only would not be the kernel's arithmetic: rebuild the kernel's exact problem for point 0 in float64
plane = (D // 2) * (D // 2)
Vcat = torch.cat([V1[0], V2[0]], dim=1)                                # [R, 2C]


def shifted(X, kd):                                                     # rows r + (kd - 1) plane, zero outside the scene's depth range
    out = torch.zeros_like(X)
    s = (kd - 1) * plane
    if s == 0:
        return X.clone()
    if s > 0:
        out[:-s] = X[s:]
    else:
        out[-s:] = X[:s]
    return out


A = torch.cat([shifted(Vcat, kd) for kd in range(3)], dim=1)            # [R, 3 x 2C]  (K = 768)
Bm = torch.cat([U[0, kd] for kd in range(3)], dim=1).t().contiguous()    # [768, N]
ref = (A.double() @ Bm.double())
scale = ref.abs().max().item()
err = lambda x: ((x.double() - ref).abs().max().item() / scale, ((x.double() - ref).norm() / ref.norm()).item())
a, b = split3(A), split3(Bm)
mm = lambda x, y: x @ y
six = mm(a[0], b[0]) + (mm(a[0], b[1]) + mm(a[1], b[0])) + (mm(a[1], b[1]) + mm(a[0], b[2]) + mm(a[2], b[0]))
four = mm(a[0], b[0]) + (mm(a[0], b[1]) + mm(a[1], b[0])) + mm(a[1], b[1])
three = mm(a[0], b[0]) + (mm(a[0], b[1]) + mm(a[1], b[0]))
one = mm(a[0], b[0])
print("Winograd point GEMM [%d x %d] x [%d x %d], errors against float64 as (max-abs / max |ref|, relative L2):" % (R, A.shape[1], A.shape[1], N))
print("  exact-fp32 MFMA kernel of this repo (conv_igemm_kernel)   %.2e  %.2e" % err(Mm[0]))
print("  fp32 GEMM library (rocBLAS, same operands)                %.2e  %.2e" % err(A @ Bm))
print("  3 x bf16 split, 6 cross products (fp32 accumulate)        %.2e  %.2e" % err(six))
print("  3 x bf16 split, 4 cross products (2-term operands)        %.2e  %.2e" % err(four))
print("  2 x bf16 split, 3 cross products                          %.2e  %.2e" % err(three))
print("  plain bf16 operands (1 product)                           %.2e  %.2e" % err(one))
print("bandwidth side (DESIGN.md / TUNING_LOG r5): the split operands are 6 B per element instead of 4 and the six MFMAs of a k-chunk take 1 / 2.67 of the "
      "fp32 MFMA's time: a 128 x 128 tile then needs 1536 B per k for 16384 MACs -> 19.5 TB/s of L2 -> LDS traffic at 417 TF-equivalent, against 4 TB/s for the "
      "fp32 kernel at 130 TF; a 256 x 256 tile still 9.8 TB/s")
