#!/usr/bin/env python
"""Why torch's multi-tensor clip (foreach=True) still issues one mul_ per gradient in the joint step: properties of the gradients after one backward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}
ds = syn.SyntheticDataset(1.5)
loss = train.compute_all_loss_nvs(cfg, 0, sample, ds, model, {}, dev)[0]
loss.backward()


def dense(g):
    if g.numel() == 0:
        return True
    st = sorted(zip(g.stride(), g.shape))
    exp = 1
    for s, n in st:
        if n == 1:
            continue
        if s != exp:
            return False
        exp *= n
    return True


kinds = {}
for n, p in model.named_parameters():
    g = p.grad
    if g is None:
        continue
    key = (str(g.dtype), str(g.device), dense(g), g.is_contiguous(), g.layout == torch.strided, g.stride() == p.stride())
    kinds.setdefault(key, []).append(n)
for k, v in kinds.items():
    print(k, len(v), v[:4])
grads = [p.grad for p in model.parameters() if p.grad is not None and p.grad.dtype == torch.float32]
c = torch.tensor(0.5, device=dev)
for name, fn in (("_foreach_mul_(grads, 0-dim tensor)", lambda: torch._foreach_mul_(grads, c)), ("_foreach_mul_(grads, float)", lambda: torch._foreach_mul_(grads, 0.5)),
                 ("_foreach_norm", lambda: torch._foreach_norm(grads, 2.0)), ("clip_grad_norm_ foreach", lambda: train.clip_grad_norm_(model.parameters(), 10.0))):
    fn()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p_:
        fn()
        torch.cuda.synchronize()
    ev = sorted([e for e in p_.key_averages() if e.self_device_time_total > 0], key=lambda e: -e.count)
    print("%-36s %s" % (name, ", ".join("%s x%d (%.2f ms)" % (e.key[:50], e.count, e.self_device_time_total / 1e3) for e in ev[:4])))
