"""Winograd vs direct data gradient of a non-square 3x3x3 convolution against float64 autograd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); from _variants import apply_environ  # noqa: E402,E702

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(13)
for (n, D, H, W, Ci, Co) in ((1, 16, 16, 16, 256, 128), (1, 16, 16, 16, 128, 256), (2, 4, 6, 8, 256, 128)):
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (27 * Ci) ** 0.5
    dy = torch.randn(n, D, H, W, Co, generator=g)
    x = torch.zeros(n, Ci, D, H, W, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x, w.double(), padding=1).backward(dy.double().permute(0, 4, 1, 2, 3))
    ref = x.grad.permute(0, 2, 3, 4, 1)
    wp = co.pack_conv3d_weight(w.to(dev))
    for mode in ("0", "1"):
        os.environ["FORGE_WINOGRAD"] = mode
        apply_environ()
        dx = torch.empty(n, D, H, W, Ci, device=dev)
        co.conv3_launch(dy.to(dev), Co, None, 0, wp, None, dx, (n, D, H, W), Ci, dgrad=True)
        e = dx.double().cpu() - ref
        print((n, D, H, W, Ci, Co), "mode", mode, "applies", co.wino_applies(co.TAPS_3x3x3, 1, n, D, H, W, Co, 0, Ci),
              "max err %.3e  rel L2 %.3e" % (e.abs().max().item(), (e.norm() / ref.norm()).item()))
