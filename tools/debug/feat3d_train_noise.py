"""Encoder3D.get_feat3D in TRAIN mode vs the oracle autograd: relative max error of the output and of a few parameter gradients, on the
direct kernels and with the Winograd launches (conv1) - how much of the 1e-2-level gradient disagreement is the 53 train-mode
BatchNorm layers amplifying fp32 reordering noise, whichever kernel computes the convolutions."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import forge_oracle as fo  # noqa: E402
from forge_amd import synthetic as syn  # noqa: E402
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); from _variants import apply_environ  # noqa: E402,E702
from forge_amd.encoder import Encoder3D  # noqa: E402

dev = torch.device("cuda:0")
names = ("conv1.0.weight", "feature_extraction.7.2.conv3.weight", "feature_extraction.6.0.conv2.weight", "feature_extraction.5.0.downsample.0.weight",
         "feature_extraction.4.0.conv1.weight", "feature_extraction.0.weight", "feature_extraction.7.0.bn2.weight")
for seed in (31, 41):
    enc = Encoder3D(syn.kubric_config())
    w = syn.seeded_state_dict({"encoder_3d." + k: v for k, v in enc.state_dict().items()}, 0)
    enc.load_state_dict({k[len("encoder_3d."):]: v for k, v in w.items()})
    enc = enc.to(dev).train()
    img = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(seed))
    gy = None
    res = {}
    for mode in ("ref", "ref64", "0", "1"):
        if mode.startswith("ref"):
            dt = torch.float64 if mode == "ref64" else torch.float32
            wr = {k: v.clone().to(dt if v.dtype.is_floating_point else v.dtype).requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in w.items()}
            out = fo.get_feat3D(img.to(dt), wr, training=True)
            if gy is None:
                gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed + 1))
            out.backward(gy.to(dt))
            res[mode] = (out.detach().double(), {n: wr["encoder_3d." + n].grad.double() for n in names})
        else:
            os.environ["FORGE_WINOGRAD"] = mode
            apply_environ()
            enc.zero_grad(set_to_none=True)
            got = enc.get_feat3D(img.to(dev))
            got.backward(gy.to(dev))
            p = dict(enc.named_parameters())
            res[mode] = (got.detach().double().cpu(), {n: p[n].grad.double().cpu() for n in names})
    rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
    for a in ("ref", "0", "1"):
        print("seed %d  %-5s vs float64 oracle: out %.2e | " % (seed, {"ref": "fp32 oracle", "0": "direct", "1": "winograd"}[a], rel(res[a][0], res["ref64"][0]))
              + "  ".join("%.1e" % rel(res[a][1][n], res["ref64"][1][n]) for n in names))
