#!/usr/bin/env python
"""Micro-benchmark of forge_conv_wgrad on ResNet-like 2-D shapes (small M) and the 3-D ConvGRU shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
T1 = [(0, 0, 0)]
T9 = [(0, a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]
cases = [("l4 1x1 512->2048", 5, 32, 32, 512, 2048, T1), ("l4 1x1 2048->512", 5, 32, 32, 2048, 512, T1), ("l4 3x3 512->512", 5, 32, 32, 512, 512, T9),
         ("l3 3x3 256->256", 5, 32, 32, 256, 256, T9), ("l3 1x1 1024->256", 5, 32, 32, 1024, 256, T1), ("l1 3x3 64->64", 5, 64, 64, 64, 64, T9),
         ("l1 1x1 64->256", 5, 64, 64, 64, 256, T1)]
for name, n, H, W, Cin, Cout, taps in cases:
    M = n * H * W
    x = torch.randn(n, 1, H, W, Cin, device=dev)
    dy = torch.randn(n, 1, H, W, Cout, device=dev)
    dwp = torch.zeros(len(taps), Cout, Cin, device=dev)
    f = lambda: co.conv_wgrad(dy, x, Cin, None, 0, dwp, (n, 1, H, W), (1, H, W), Cout, taps)
    f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-20s M=%-6d %.3f ms  %.1f TF" % (name, M, ms, 2.0 * M * Cout * Cin * len(taps) / ms / 1e9))
