import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from forge_amd import convops as co
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); from _variants import apply_environ  # noqa: E402,E702
dev = torch.device("cuda:0")
rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
g = torch.Generator().manual_seed(3)
wct = (torch.randn(128, 64, 4, 4, 4, generator=g) / 40).to(dev)
wT = wct.reshape(128, 64, 64).permute(2, 0, 1).contiguous()
taps = [(kz - 1, ky - 1, kx - 1) for kz in range(4) for ky in range(4) for kx in range(4)]
for D in (8, 16, 32):
    D2 = 2 * D
    gu = torch.randn(1, D2, D2, D2, 64, generator=g).to(dev)
    ref = F.conv3d(gu.permute(0, 4, 1, 2, 3).contiguous(), wct, stride=2, padding=1).permute(0, 2, 3, 4, 1)
    for tile in ("", "A", "B", "C", "D"):
        for ks in ("", "1", "2", "4"):
            os.environ.pop("FORGE_CONV_TILE", None); os.environ.pop("FORGE_CONV_KSPLIT", None)
            apply_environ()
            if tile: os.environ["FORGE_CONV_TILE"] = tile
            apply_environ()
            if ks: os.environ["FORGE_CONV_KSPLIT"] = ks
            apply_environ()
            dz = torch.full((1, D, D, D, 128), float("nan"), device=dev)
            try:
                co.conv_igemm(gu, 64, 64, None, 0, 0, wT, None, None, None, 1.0, None, None, None, dz, None, (1, D, D, D), (D2, D2, D2), 128, 128, taps, istride=2,
                              epilogue=co.EPI_BIAS)
                plan = co.conv_plan(D ** 3, 128, 64, 64, co.EPI_BIAS, 128)
                print("D=%d tile=%-1s ksplit=%-1s plan=%s  rel=%.2e" % (D, tile or "-", ks or "-", plan, rel(dz, ref)))
            except Exception as e:
                print("D=%d tile=%s ks=%s: %s" % (D, tile, ks, str(e)[:80]))
