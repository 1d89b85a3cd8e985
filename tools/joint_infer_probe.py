#!/usr/bin/env python
"""FORGE with PREDICTED poses in inference (kubric_eval.py predict_initial / demo.py: 5 input views -> poses from the 2-D + 3-D estimators -> reconstruction ->
10 rendered views): eager and hipGraph replay, next to the GT-pose forward of the same model class."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.graph import GraphedForward  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

dev = torch.device("cuda:0")
ds = syn.SyntheticDataset(1.5)
sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


TRACE_STEPS = int(os.environ.get("JOINT_INFER_TRACE", "0"))     # > 0: rocprofv3 run - that many eager predicted-pose forwards between two marker kernels

for gt in ((False,) if TRACE_STEPS else (False, True)):
    cfg = syn.kubric_config(use_gt_pose=gt, parameter="joint")
    model = FORGE(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).eval()

    def eager():
        with torch.no_grad():
            return model(sample, ds, dev)
    if TRACE_STEPS:
        for _ in range(3):
            eager()
        torch.cuda.synchronize()
        torch.cuda._sleep(1000)
        for _ in range(TRACE_STEPS):
            eager()
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        print("joint inference: traced %d eager forwards" % TRACE_STEPS)
        break
    e = timed(eager)
    try:
        g = GraphedForward(model, sample, ds, dev)
        r = timed(lambda: g(sample))
    except Exception as ex:
        r = repr(ex)[:200]
    if not gt:
        feats = torch.randn(1, 5, 128, 32, 32, 32, device=dev) * 0.5
        clips = sample["images"][:, :5].contiguous()
        with torch.no_grad():
            t3 = timed(lambda: model.encoder_traj(feats, return_features=True))
            t2 = timed(lambda: model.encoder_traj_2d(clips, return_features=True))
            import stock_pose                                         # tools/stock_pose.py: the same modules on torch's own kernels
            with stock_pose.patched(model.encoder_traj, model.encoder_traj_2d):
                s3 = timed(lambda: model.encoder_traj(feats, return_features=True))
                s2 = timed(lambda: model.encoder_traj_2d(clips, return_features=True))
        print("  alone, eager: 3-D pose estimator %.2f ms (stock torch %.2f), 2-D pose estimator %.2f ms (stock torch %.2f)" % (t3, s3, t2, s2))
    print("FORGE inference, %s poses, 10 rendered views: eager %.2f ms, hipGraph replay %s" % ("GT" if gt else "predicted", e, r if isinstance(r, str) else "%.2f ms" % r))
