#!/usr/bin/env python
"""cProfile of the HOST side of the eager joint step (where do the ~60 ms of Python / launch time per step go?): top functions by own time."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import synthetic as syn, train  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

dev = torch.device("cuda:0")
cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
cfg.loss.regu_origin_proj = 1.0
model = FORGE(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
params = [p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head, model.render) for p in m.parameters()]
opt = torch.optim.Adam(params, lr=1e-4, fused=True)
sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}
ds = syn.SyntheticDataset(1.5)


def step():
    loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, sample, ds, model, {}, dev)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    train.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
