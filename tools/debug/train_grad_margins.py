#!/usr/bin/env python
"""Margins of the training-gradient parity tests: per key, max |g_hip - g_ref| / max |g_ref| and 1 - cosine against the reference's golden
gradients (tests/golden/train_pose3d.npz), for TWO runs of the HIP step, plus the run-to-run distance of the HIP step itself."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

dev = torch.device("cuda:0")
gold = np.load(os.path.join(ROOT, "tests", "golden", "train_pose3d.npz"))
cfg = syn.kubric_config()


def run():
    model = FORGE_poseEstimator3D(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), int(gold["weight_seed"])))
    model = model.to(dev).train()
    sample = syn.make_sample(1, 5, 256, 1.5, seed=int(gold["sample_seed"]))
    tgt_i = sample["images"][0].repeat(2, 1, 1, 1).to(dev)
    tgt_m = sample["fg_probabilities"][0].repeat(2, 1, 1, 1).to(dev)
    imgs, masks = model(sample, syn.SyntheticDataset(1.5), dev)
    loss = 5.0 * torch.nn.functional.mse_loss(imgs, tgt_i) + torch.nn.functional.mse_loss(masks, tgt_m)
    loss.backward()
    return float(loss), {k: v.grad.detach().cpu() for k, v in model.named_parameters() if v.grad is not None}


la, ga = run()
lb, gb = run()
print("loss a %.9f  b %.9f  golden %.9f" % (la, lb, float(gold["loss"])))
keys = [k[len("grad__"):] for k in gold.files if k.startswith("grad__")]
worst = [0.0, 0.0, 0.0]
for k in keys:
    ref = torch.from_numpy(gold["grad__" + k])
    e = (ga[k] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    cos = torch.nn.functional.cosine_similarity(ga[k].double().flatten(), ref.double().flatten(), dim=0).item() if ref.numel() > 1 else 1.0
    rr = (ga[k] - gb[k]).abs().max().item() / max(ga[k].abs().max().item(), 1e-30)
    worst = [max(worst[0], e), max(worst[1], 1 - cos), max(worst[2], rr)]
    print("%-62s |g|max %.3e  vs golden %.2e  1-cos %.2e  run-to-run %.2e" % (k, ref.abs().max().item(), e, 1 - cos, rr))
allrr = max((ga[k] - gb[k]).abs().max().item() / max(ga[k].abs().max().item(), 1e-30) for k in ga)
print("worst: vs golden %.2e  1-cos %.2e  run-to-run (golden keys) %.2e  run-to-run (all %d parameters) %.2e  bit-identical %s"
      % (worst[0], worst[1], worst[2], len(ga), allrr, all(torch.equal(ga[k], gb[k]) for k in ga)))
