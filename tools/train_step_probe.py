#!/usr/bin/env python
"""Time one GT-pose training step (BASELINE configs[3] shape per GPU: FORGE_poseEstimator3D, b scenes x 5 views -> 10 rendered
views/scene, MSE rgb+mask (the fused squared-error pass of forge_amd.train), grad-clip 10, Adam - scripts/kubric_trainer.py:51-59) on one MI355X.
TRAIN_GRID=64: the 128^3-voxel scenes of configs[3] - synthetic [b,5,128,64^3] feature volumes (the encoder cannot produce them from
256^2 images) through FORGE_poseEstimator3D.reconstruct: rotate(D=64), three fusions, heads to 128^3, 10 ray-marched views, backward, Adam."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from forge_amd import synthetic as syn, train  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

b = int(os.environ.get("TRAIN_SCENES", "1"))
steps = int(os.environ.get("TRAIN_STEPS", "5"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE_poseEstimator3D(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
use_graph = os.environ.get("TRAIN_GRAPH", "0") == "1"       # capture fwd + bwd + clip + Adam into one hipGraph (forge_amd.graph.GraphedStep)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=use_graph, fused=not use_graph)
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=3).items()}
ds = syn.SyntheticDataset(1.5)
from forge_amd import geo_utils  # noqa: E402
from forge_amd.train import grouped_mse  # noqa: E402
grid = int(os.environ.get("TRAIN_GRID", "32"))
if grid == 64:
    feats = (torch.randn(b, 5, 128, 64, 64, 64, device=dev) * 0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    cams = geo_utils.camera_dict(sample["cam_extrinsics_cv2_canonicalized"].repeat(1, 2, 1, 1), sample["K_cv2"].repeat(1, 2, 1, 1))
    run_model = lambda: model.reconstruct(feats, sample["cam_poses_cv2_canonicalized"], cams)[:2]
else:
    run_model = lambda: model(sample, ds, dev)


def loss_of(imgs, masks):
    mi = grouped_mse(imgs.reshape(b, 10, 3, 256, 256), sample["images"], 5)
    mm = grouped_mse(masks.reshape(b, 10, 1, 256, 256), sample["fg_probabilities"], 5)
    return 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]


def step():
    imgs, masks = run_model()
    loss = loss_of(imgs, masks)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    train.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    return loss


if use_graph:
    from forge_amd.graph import GraphedStep

    def graph_fn():
        imgs, masks = run_model()
        loss = loss_of(imgs, masks)
        loss.backward()
        train.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return loss.detach()
    step = GraphedStep(graph_fn, opt, warmup=2)
for _ in range(2):
    l = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("train step grid=%d^3 b=%d%s: %.1f ms/step, %.1f rendered views/s (fwd+bwd+Adam), loss %.5f, peak mem %.1f GB"
      % (2 * grid, b, " (hipGraph replay)" if use_graph else "", dt * 1e3, b * 10 / dt, l.item(), torch.cuda.max_memory_allocated() / 2 ** 30))

if os.environ.get("TRAIN_PROFILE"):
    # per-launch HIP-event timing of every conv GEMM / wgrad launch of one step, grouped by shape
    from forge_amd import convops as co
    rec = []

    def timed(kind, fn, label_of):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            rec.append((kind,) + label_of(*a, **k) + (e0, e1))
            return r
        return wrapper

    def lab_igemm(in1, C1, ld1, in2, C2, ld2, wp, *rest, **k):
        grid, in_grid, Cout = rest[9], rest[10], rest[11]
        M = grid[0] * grid[1] * grid[2] * grid[3]
        return (M, Cout, C1 + C2, wp.shape[0], k.get("istride", 1), k.get("ostride", 1))

    def lab_wgrad(dy, x1, C1, x2, C2, dwp, grid, in_grid, Cout, taps, istride=1, **k):
        M = grid[0] * grid[1] * grid[2] * grid[3]
        return (M, Cout, C1 + C2, len(taps), istride, 1)

    co.conv_igemm = timed("igemm", co.conv_igemm, lab_igemm)
    co.conv_wgrad = timed("wgrad", co.conv_wgrad, lab_wgrad)
    import forge_amd.encoder as _e, forge_amd.fusion as _f, forge_amd.volume_render as _v
    for mod in (_e, _f, _v):
        for name in ("conv_igemm", "conv_wgrad"):
            if hasattr(mod, name):
                setattr(mod, name, getattr(co, name))
    step()
    torch.cuda.synchronize()
    agg = {}
    for kind, M, N, K, T, is_, os_, e0, e1 in rec:
        key = (kind, M, N, K, T, is_, os_)
        ms = e0.elapsed_time(e1)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += ms
    tot = {"igemm": 0.0, "wgrad": 0.0}
    print("%-6s %8s %5s %5s %4s %2s %2s %5s %9s %8s" % ("kind", "M", "N", "Cin", "taps", "is", "os", "calls", "ms total", "TF"))
    for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, M, N, K, T, is_, os_ = key
        tf = 2.0 * M * N * K * T * n / (ms * 1e-3) / 1e12
        tot[kind] += ms
        print("%-6s %8d %5d %5d %4d %2d %2d %5d %9.3f %8.1f" % (kind, M, N, K, T, is_, os_, n, ms, tf))
    print("totals (event-timed, eager, incl. launch gaps):", tot)
