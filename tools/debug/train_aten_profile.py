"""aten-level view of ONE eager GT-pose training step (tools/train_step_probe.py's step): which torch ops surround the hand-written kernels, by launches
and device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from forge_amd import synthetic as syn
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
from forge_amd.train import grouped_mse

b = int(os.environ.get("TRAIN_SCENES", "1"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE_poseEstimator3D(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=3).items()}
ds = syn.SyntheticDataset(1.5)


def step():
    imgs, masks = model(sample, ds, dev)
    mi = grouped_mse(imgs.reshape(b, 10, 3, 256, 256), sample["images"], 5)
    mm = grouped_mse(masks.reshape(b, 10, 1, 256, 256), sample["fg_probabilities"], 5)
    loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(os.environ.get("ATEN_STACKS"))) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0 and (e.key.startswith("aten::") or e.key.startswith("Optimizer") or "Memcpy" in e.key or "Memset" in e.key)]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("aten / memcpy device time of one step: %.2f ms in %d launches" % (tot / 1e3, sum(e.count for e in rows)))
for e in rows[:40]:
    print("%-50s %5d calls %9.1f us  avg %6.1f us" % (e.key[:50], e.count, e.self_device_time_total, e.self_device_time_total / e.count))

if os.environ.get("ATEN_STACKS"):      # where the small ops come from: python frames under forge_amd / torch.optim / torch.nn.utils per op
    import collections
    want = set(os.environ["ATEN_STACKS"].split(","))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.name in want and e.device_time_total > 0:
            frames = [f for f in (e.stack or []) if ("forge_amd" in f or "optim" in f or "clip_grad" in f or "tools/" in f)]
            agg[(e.name, frames[0] if frames else "(autograd engine / no python frame)")][0] += 1
            agg[(e.name, frames[0] if frames else "(autograd engine / no python frame)")][1] += e.device_time_total
    for (name, fr), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-14s %4d calls %8.1f us  %s" % (name, n, us, fr[-110:]))
