#!/usr/bin/env python
"""Launch forge_conv_wgrad (ConvGRU gates shape) a few times — target of rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402

dev = torch.device("cuda:0")
D, Cc = 32, 128
M = D ** 3
x, hbuf = torch.randn(M, Cc, device=dev), torch.randn(M, Cc, device=dev)
dy = torch.randn(M, 256, device=dev)
dwp = torch.zeros(27, 256, 256, device=dev)
for _ in range(5):
    co.conv_wgrad(dy, x, Cc, hbuf, Cc, dwp, (1, D, D, D), (D, D, D), 256, co.TAPS_3x3x3)
torch.cuda.synchronize()
