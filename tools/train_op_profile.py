#!/usr/bin/env python
"""torch.profiler view of one GT-pose training step: which aten ops (copies, strided element-wise, reductions, BatchNorm) account for
the non-GEMM time around the hand-written kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

b = int(os.environ.get("TRAIN_SCENES", "1"))
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


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48, max_shapes_column_width=70))
