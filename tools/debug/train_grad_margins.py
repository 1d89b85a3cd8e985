#!/usr/bin/env python
"""Margins of the training-gradient parity tests against the REFERENCE's float64 evaluation of the same step (tests/golden/train_pose3d.npz
and train_joint.npz: grad64__* / gsub64__* next to the reference's fp32 gradients, oracle/make_golden.py): per key
    hip/f64   max |g_hip - g_f64| / max |g_f64| and 1 - cos       ref32/f64   the same for the reference's own fp32 run (stored)
    ratio     hip/f64 over ref32/f64                              run-to-run  two HIP steps
for the default launch path and with the Winograd launches off (MARGIN_MODES=default,direct)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))          # tests/test_gpu_configs.py imports the oracle (the checker) at module level
import numpy as np  # noqa: E402
import torch  # noqa: E402

from forge_amd import convops as co, synthetic as syn  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

dev = torch.device("cuda:0")
cfg = syn.kubric_config()
T = lambda a: torch.from_numpy(np.asarray(a))


def run_pose3d(gold):
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


def run_joint(gold):
    from test_gpu_configs import joint_training_step
    c = syn.kubric_config(use_gt_pose=False, parameter="joint")
    c.loss.recon_rgb, c.loss.recon_mask, c.loss.regu_origin_proj = float(gold["recon_rgb"]), float(gold["recon_mask"]), float(gold["regu_origin_proj"])
    loss, _, model, _, _ = joint_training_step(dev, c, int(gold["weight_seed"]), int(gold["sample_seed"]))
    return float(loss), {k: v.grad.detach().cpu() for k, v in model.named_parameters() if v.grad is not None}


def dist(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item() if a.numel() > 1 else 1.0
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-300), 1.0 - cos


def table(name, gold, run):
    keys = [k[len("g64err__"):] for k in gold.files if k.startswith("g64err__")]
    la, ga = run(gold)
    lb, gb = run(gold)
    print("== %s: loss hip %.9f / %.9f   reference fp32 %.9f   float64 %.9f" % (name, la, lb, float(gold["loss"]), float(gold["loss64"])))
    worst = 0.0
    for k in keys:
        g = ga[k].flatten()
        g2 = gb[k].flatten()
        if "gsub64__" + k in gold.files:
            st = int(gold["gstride__" + k])
            g, g2, r64 = g[::st], g2[::st], T(gold["gsub64__" + k])
            e32, c32 = float(gold["g64suberr__" + k]), float(gold["g64subcos__" + k])
        else:
            r64 = T(gold["grad64__" + k]).flatten()
            e32, c32 = float(gold["g64err__" + k]), float(gold["g64cos__" + k])
        e, c = dist(g, r64)
        rr = (g - g2).abs().max().item() / max(g.abs().max().item(), 1e-30)
        noise = r64.abs().max().item() < 1e-5
        worst = max(worst, 0.0 if noise else e / max(e32, 1e-12))
        print("%-62s |g64|max %.3e  hip/f64 %.2e (1-cos %.2e)  ref32/f64 %.2e (1-cos %.2e)  ratio %5.2f  run-to-run %.1e%s"
              % (k[-62:], r64.abs().max().item(), e, c, e32, c32, e / max(e32, 1e-12), rr, "  [pure cancellation noise]" if noise else ""))
    print("worst ratio hip/f64 : ref32/f64 = %.2f" % worst)


modes = os.environ.get("MARGIN_MODES", "default,direct").split(",")
which = os.environ.get("MARGIN_STEPS", "pose3d,joint").split(",")
for mode in modes:
    print("######## launch path: %s" % mode)
    with co.winograd(mode != "direct"):
        if "pose3d" in which:
            table("GT-pose step (train_pose3d.npz)", np.load(os.path.join(ROOT, "tests", "golden", "train_pose3d.npz")), run_pose3d)
        if "joint" in which:
            table("joint step (train_joint.npz)", np.load(os.path.join(ROOT, "tests", "golden", "train_joint.npz")), run_joint)
