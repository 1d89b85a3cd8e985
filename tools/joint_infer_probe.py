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


for gt in (False, True):
    cfg = syn.kubric_config(use_gt_pose=gt, parameter="joint")
    model = FORGE(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).eval()

    def eager():
        with torch.no_grad():
            return model(sample, ds, dev)
    e = timed(eager)
    try:
        g = GraphedForward(model, sample, ds, dev)
        r = timed(lambda: g(sample))
    except Exception as ex:
        r = repr(ex)[:200]
    print("FORGE inference, %s poses, 10 rendered views: eager %.2f ms, hipGraph replay %s" % ("GT" if gt else "predicted", e, r if isinstance(r, str) else "%.2f ms" % r))
