#!/usr/bin/env python
"""Time one joint-mode training step (BASELINE configs[4] code path: FORGE with predicted poses - 2-D + 3-D pose estimators and the
pose head in stock torch, reconstruction on the HIP kernels - fwd + bwd + Adam) and list the slowest kernels of the last step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

b = int(os.environ.get("JOINT_SCENES", "1"))
steps = int(os.environ.get("JOINT_STEPS", "4"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
model = FORGE(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 10, 256, 1.5, seed=12).items()}
ds = syn.SyntheticDataset(1.5)


def step():
    imgs, masks, origin_proj, pose = model(sample, ds, dev)
    loss = F.mse_loss(imgs, sample["images"].reshape(-1, 3, 256, 256)) + F.mse_loss(masks, sample["fg_probabilities"].reshape(-1, 1, 256, 256)) \
        + F.mse_loss(pose["pred"], pose["gt"]) + 0.1 * F.mse_loss(origin_proj, torch.full_like(origin_proj, 0.5))
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
print("joint train step b=%d: %.1f ms/step (fwd+bwd+Adam, 10 rendered views/scene), loss %.5f, peak mem %.1f GB"
      % (b, dt * 1e3, l.item(), torch.cuda.max_memory_allocated() / 2 ** 30))
