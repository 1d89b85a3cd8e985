#!/usr/bin/env python
"""torch.profiler view of one joint 2D3D fine-tune step (tools/joint_step_probe.py's step): which aten ops account for the stock-torch share
(element-wise, copies, reductions, GEMMs of the attention blocks) around the libforge kernels; grouped by op and input shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

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
# phase by phase: which part of the step issues the ~550-per-step element-wise launches
def phase(name, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p_:
        out = fn()
        torch.cuda.synchronize()
    ev = [e for e in p_.key_averages() if e.key.startswith("aten::") and e.self_device_time_total > 0]
    ev.sort(key=lambda e: -e.self_device_time_total)
    print("%-10s %s" % (name, ", ".join("%s x%d (%.2f ms)" % (e.key, e.count, e.self_device_time_total / 1e3) for e in ev[:14])))
    return out


loss = phase("forward", lambda: train.compute_all_loss_nvs(cfg, 0, sample, ds, model, {}, dev)[0])
opt.zero_grad(set_to_none=True)
phase("backward", lambda: loss.backward())
phase("clip", lambda: train.clip_grad_norm_(model.parameters(), 10.0))
phase("adam", lambda: opt.step())
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=44, max_shapes_column_width=80))
