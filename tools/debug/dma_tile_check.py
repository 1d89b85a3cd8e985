"""Every tile of conv_igemm_kernel (LDS-DMA staged K loop) on ragged direct and Winograd problems: all tiles must agree to fp32 summation-order
noise; then the time of the gates launches (tools/debug/gemm_ceiling.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from forge_amd import convops as co
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for (B, D, C1, C2, Cout, taps) in ((1, 8, 32, 0, 32, co.TAPS_3x3x3), (1, 16, 128, 128, 256, co.TAPS_3x3x3), (2, 12, 64, 32, 96, co.TAPS_3x3x3), (1, 10, 32, 0, 40, ((0, 0, 0),))):
    M = B * D ** 3
    x = torch.randn(M, C1, device=dev)
    h = torch.randn(M, C2, device=dev) if C2 else None
    w = torch.randn(len(taps), Cout, C1 + C2, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    outs = {}
    for tile in os.environ.get("CHECK_TILES", "ABCDE"):
        for ks in (1, 3):
            if ks > len(taps) * ((C1 + C2) // 32):
                continue
            o = torch.empty(M, Cout, device=dev)
            with co.force_plan(tile=tile, ksplit=ks):
                co.conv_igemm(x, C1, C1, h, C2, C2, w, bias, None, None, 1.0, None, None, None, o, None, (B, D, D, D), (D, D, D), Cout, Cout, taps, epilogue=co.EPI_BIAS)
            outs[(tile, ks)] = o
    torch.cuda.synchronize()
    ref = outs[("A", 1)]
    for k, o in outs.items():
        err = (o - ref).abs().max().item()
        ok = err < 2e-5 * ref.abs().max().item()
        bad += not ok
        print("shape", (B, D, C1, C2, Cout, len(taps)), k, "max diff vs A/1 %.2e" % err, "" if ok else "MISMATCH")
print("dma_tile_check:", "OK" if not bad else "%d MISMATCHES" % bad)
