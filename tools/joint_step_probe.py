#!/usr/bin/env python
"""Time the joint 2D3D fine-tune iteration (BASELINE configs[4]; kubric_train_joint.py:111-141) exactly as bench.py's extra_configs
`joint_step` / `joint_step_grid64` run it (bench.joint_configs), for rocprofv3 passes:

    JOINT_GRID=32|64  JOINT_STEPS=n  [JOINT_SCENES=b]  [JOINT_STOCK=1]   python tools/joint_step_probe.py

FORGE with predicted poses (2-D + 3-D pose estimators and the pose head on stock torch kernels; everything else on libforge_hip.so),
compute_all_loss_nvs, backward, clip 10, Adam over the reference's parameter list."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import synthetic as syn, train  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

grid = int(os.environ.get("JOINT_GRID", "32"))
scenes = int(os.environ.get("JOINT_SCENES", "1"))
steps = int(os.environ.get("JOINT_STEPS", "4"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
cfg.loss.regu_origin_proj = 1.0
model = FORGE(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
if os.environ.get("JOINT_STOCK") == "1":       # the round-4 state: both pose estimators entirely on stock torch kernels (MIOpen / rocBLAS)
    import stock_pose                           # tools/stock_pose.py (this script's own directory is on sys.path)
    for _m in (model.encoder_traj, model.encoder_traj_2d):
        _m.forward = stock_pose.stock_forward(_m)
params = [p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head, model.render) for p in m.parameters()]
opt = torch.optim.Adam(params, lr=1e-4, fused=True)
sample = {k: v.to(dev) for k, v in syn.make_sample(scenes, 10, 256, 1.5, seed=12).items()}
ds = syn.SyntheticDataset(1.5)
call = model
if grid == 64:
    gen = torch.Generator(device=dev).manual_seed(79)
    f64 = torch.randn(scenes, 5, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    call = lambda s, d, dv: model(s, d, dv, features_recon=f64)      # noqa: E731


def step():
    loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, sample, ds, call, {}, dev)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    train.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    return loss


for _ in range(2):
    l = step()
torch.cuda.synchronize()
torch.cuda._sleep(1000)                    # marker kernel (`spin_kernel`) for tools/joint_kernel_share.py: the timed steps start here
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
torch.cuda._sleep(1000)                    # ... and end here
torch.cuda.synchronize()
print("joint step grid %d, %d scene(s): %.1f ms/step (fwd+bwd+clip+Adam, 10 rendered views), loss %.5f, peak mem %.1f GB, steps timed %d (+2 warm-up)"
      % (grid, scenes, dt * 1e3, l.item(), torch.cuda.max_memory_allocated() / 2 ** 30, steps))
