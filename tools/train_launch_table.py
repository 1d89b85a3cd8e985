#!/usr/bin/env python
"""Every matrix-core launch of ONE GT-pose training step (TRAIN_SCENES scenes, default 4; TRAIN_MODE=joint: the joint 2D3D fine-tune step of
BASELINE configs[4], 1 scene; TRAIN_MODE=infer / infer_pose3d: the inference forward of FORGE / FORGE_poseEstimator3D) with HIP events around it: entry point, shape
(rows, Cout, Cin, taps / kd), ms, TFLOP/s of the FLOPs it executes - grouped by shape, sorted by time. Shows which GEMM shapes sit
furthest below the 157.3 TF pipe."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import _lib, flopmeter as fmod, synthetic as syn  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402
from forge_amd.train import grouped_mse  # noqa: E402

b = int(os.environ.get("TRAIN_SCENES", "4"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE_poseEstimator3D(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=3).items()}
ds = syn.SyntheticDataset(1.5)


mode = os.environ.get("TRAIN_MODE", "gt_pose")
if mode == "joint":
    from forge_amd import train
    from forge_amd.model import FORGE
    b = 1
    cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
    cfg.loss.regu_origin_proj = 1.0
    model = FORGE(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).train()
    opt = torch.optim.Adam([p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head, model.render)
                            for p in m.parameters()], lr=1e-4, fused=True)
    sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}


if mode in ("infer", "infer_pose3d"):
    # the headline forward (FORGE, GT poses, 5 in / 5 out) or FORGE_poseEstimator3D inference (10 rendered views), eval / no_grad, TRAIN_SCENES scenes
    from forge_amd.model import FORGE
    cls = FORGE if mode == "infer" else FORGE_poseEstimator3D
    model = cls(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).eval()


def step():
    if mode in ("infer", "infer_pose3d"):
        with torch.no_grad():
            model(sample, ds, dev)
        return
    if mode == "joint":
        loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, sample, ds, model, {}, dev)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        train.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return
    imgs, masks = model(sample, ds, dev)
    mi = grouped_mse(imgs.reshape(b, 10, 3, 256, 256), sample["images"], 5)
    mm = grouped_mse(masks.reshape(b, 10, 1, 256, 256), sample["fg_probabilities"], 5)
    loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
v = fmod._v
shape_of = {
    "forge_conv_igemm": lambda a: (v(a[19]) * v(a[20]) * v(a[21]) * v(a[22]), v(a[27]), v(a[1]) + v(a[5]), v(a[30]), "is%d os%d" % (v(a[23]), v(a[31]))),
    "forge_wino_gemm": lambda a: (v(a[12]) * v(a[13]) * v(a[14]) * v(a[15]), v(a[16]), v(a[1]) + v(a[6]), v(a[17]), "16 points"),
    "forge_wino_gemm_half": lambda a: (v(a[12]) * v(a[13]) * v(a[14]) * v(a[15]), v(a[16]), v(a[1]) + v(a[6]), v(a[17]), "16 points, 8 planes out"),
    "forge_conv_wgrad": lambda a: (v(a[11]) * v(a[12]) * v(a[13]) * v(a[14]), v(a[19]), v(a[3]) + v(a[7]), v(a[21]), "is%d" % v(a[15])),
    "forge_wino_wgrad": lambda a: (v(a[10]) * v(a[11]) * v(a[12]) * v(a[13]), v(a[14]), v(a[2]) + v(a[6]), v(a[15]), "16 points"),
    "forge_attention_fwd": lambda a: (v(a[5]) * v(a[6]), v(a[7]), v(a[8]), 1, "keys x d"),
}
rec = []
L = _lib.lib()
orig = {}
for name, fl in fmod._ENTRIES.items():
    o = getattr(L, name)
    orig[name] = o

    def wrapped(*a, _o=o, _fl=fl, _n=name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = _o(*a)
        e1.record()
        rec.append((_n, shape_of[_n](a), _fl(a), e0, e1))
        return r
    setattr(L, name, wrapped)
step()
torch.cuda.synchronize()
for name, o in orig.items():
    setattr(L, name, o)
agg = collections.OrderedDict()
for n, sh, fl, e0, e1 in rec:
    a = agg.setdefault((n, sh), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += fl
tot_ms, tot_fl = sum(a[1] for a in agg.values()), sum(a[2] for a in agg.values())
print("%d scene(s): %d matrix-core launches, %.2f ms (events, incl. launch gaps), %.1f GFLOP executed -> %.1f TF; floor at 157.3 TF: %.2f ms"
      % (b, len(rec), tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9, tot_fl / 157.3e9))
print("%-18s %9s %5s %5s %4s %-10s %5s %9s %7s %6s" % ("entry", "rows", "Cout", "Cin", "taps", "", "calls", "ms total", "TF", "lost ms"))
for (n, sh), (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-18s %9d %5d %5d %4d %-10s %5d %9.3f %7.1f %6.2f" % (n.replace("forge_", ""), sh[0], sh[1], sh[2], sh[3], sh[4], c, ms, fl / ms / 1e9, ms - fl / 130e9))
