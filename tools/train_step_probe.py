#!/usr/bin/env python
"""Time one GT-pose training step (BASELINE configs[3] shape per GPU: FORGE_poseEstimator3D, b scenes x 5 views -> 10 rendered
views/scene, MSE rgb+mask, grad-clip 10, Adam — scripts/kubric_trainer.py:51-59) on one MI355X."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

b = int(os.environ.get("TRAIN_SCENES", "1"))
steps = int(os.environ.get("TRAIN_STEPS", "5"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE_poseEstimator3D(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=3).items()}
ds = syn.SyntheticDataset(1.5)
tgt_i = sample["images"].repeat(1, 2, 1, 1, 1).reshape(-1, 3, 256, 256)
tgt_m = sample["fg_probabilities"].repeat(1, 2, 1, 1, 1).reshape(-1, 1, 256, 256)


def step():
    imgs, masks = model(sample, ds, dev)
    loss = 5.0 * F.mse_loss(imgs, tgt_i) + F.mse_loss(masks, tgt_m)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    return loss


for _ in range(2):
    l = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("train step b=%d: %.1f ms/step, %.1f rendered views/s (fwd+bwd+Adam), loss %.5f, peak mem %.1f GB"
      % (b, dt * 1e3, b * 10 / dt, l.item(), torch.cuda.max_memory_allocated() / 2 ** 30))
