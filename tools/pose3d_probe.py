#!/usr/bin/env python
"""FORGE_poseEstimator3D inference (eval, GT poses): 5 input views -> three fusions -> 10 rendered views per scene. Times the forward
with the shared input halves of the GRU convolutions (Encoder3D.fuse_groups -> ConvGRU_3D.fuse_groups_hip) against three
independent fusions, eager and as a hipGraph replay.  POSE3D_SCENES=b"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.graph import GraphedForward  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

b = int(os.environ.get("POSE3D_SCENES", "1"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE_poseEstimator3D(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).eval()
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=3).items()}
ds = syn.SyntheticDataset(1.5)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {}
for name in ("shared", "separate"):
    if name == "separate":
        enc = model.encoder_3d
        enc.fuse_groups = lambda x, groups: [enc.fuse(enc._views(x, g)) for g in groups]
    g = GraphedForward(model, sample, ds, dev)
    out = g(sample)[0].clone()
    res[name] = (timeit(lambda: g(sample)), out)
    print("pose3d inference b=%d, %s input halves: %.2f ms per forward, %.1f rendered views/s (hipGraph replay)" % (b, name, res[name][0] * 1e3, b * 10 / res[name][0]))
print("max |shared - separate| on the rendered rgb: %.2e" % (res["shared"][1] - res["separate"][1]).abs().max().item())
