#!/usr/bin/env python
"""Eval-mode BatchNorm under autograd: forge_bn_eval_fwd + sync-backward kernels (fusion._BNEvalRows) vs the torch module on the GPU vs float64 on the CPU."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd.fusion import bn_act_rows  # noqa: E402

dev = torch.device("cuda:0")
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
for shape, slope, with_res in (((2, 8, 8, 8, 256), 0.01, False), ((5, 32, 32, 2048), 0.0, True), ((5, 32, 32, 32, 128), 0.01, False), ((3, 16, 20, 64), 1.0, False)):
    torch.manual_seed(1)
    C, nd = shape[-1], len(shape)
    bn = (torch.nn.BatchNorm3d if nd == 5 else torch.nn.BatchNorm2d)(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5)
        bn.bias.copy_(torch.randn(C) * 0.3)
        bn.running_mean.copy_(torch.randn(C) * 0.1)
        bn.running_var.copy_(torch.rand(C) + 0.5)
    bn.eval()
    x, res, dy = torch.randn(*shape) * 1.3 + 0.2, torch.randn(*shape), torch.randn(*shape)
    act = (lambda t: t) if slope == 1.0 else ((lambda t: torch.relu(t)) if slope == 0.0 else (lambda t: torch.nn.functional.leaky_relu(t, slope)))
    out = {}
    for tag, mod, cast in (("f64", copy.deepcopy(bn).double(), lambda t: t.double()), ("torch-gpu", copy.deepcopy(bn).to(dev), lambda t: t.to(dev)), ("hip", copy.deepcopy(bn).to(dev), lambda t: t.to(dev))):
        xi, ri = cast(x).requires_grad_(True), cast(res).requires_grad_(True)
        if tag == "hip":
            y = bn_act_rows(mod, xi, slope, residual=ri if with_res else None)
        else:
            y = mod(xi.permute(0, nd - 1, *range(1, nd - 1))).permute(0, *range(2, nd), 1)
            y = act(y + ri if with_res else y)
        y.backward(cast(dy))
        out[tag] = (y.detach(), xi.grad, mod.weight.grad, mod.bias.grad)
    for i, name in enumerate(("y", "dx", "dgamma", "dbeta")):
        print("%-22s slope %-4s res %-5s %-7s hip/f64 %.2e   torch-gpu/f64 %.2e   hip/torch-gpu %.2e" % (
            shape, slope, with_res, name, rel(out["hip"][i], out["f64"][i]), rel(out["torch-gpu"][i], out["f64"][i]), rel(out["hip"][i], out["torch-gpu"][i])))
