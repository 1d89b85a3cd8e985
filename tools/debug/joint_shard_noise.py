#!/usr/bin/env python
"""Noise floor of the joint fine-tune step's parameter gradients: two single-process runs against each other, and (2 ranks sharing the GPU)
the ray-sharded step against the single-process one - per parameter key, relative L2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_ddp as T  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    if os.environ.get("JOINT_SPAWN_FIRST") == "1":          # the test's order: the two ranks first, then this process
        res = T._spawn2(T._joint_worker, timeout=600)
        la, ta, ga = T._joint_step(dev, False)
        lb, tb, gb = T._joint_step(dev, False)
    else:
        la, ta, ga = T._joint_step(dev, False)
        lb, tb, gb = T._joint_step(dev, False)
        res = T._spawn2(T._joint_worker, timeout=600)
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-20))
    print("loss single a/b: %.8f %.8f  sharded r0/r1: %.8f %.8f" % (la, lb, res[0][0], res[1][0]))
    for k in T.JOINT_KEYS:
        print("%-60s |g| %.3e  single-vs-single %.2e  shard0-vs-single %.2e  shard1-vs-single %.2e  shard0-vs-shard1 %.2e"
              % (k, np.linalg.norm(ga[k]), rel(gb[k], ga[k]), rel(res[0][2][k], ga[k]), rel(res[1][2][k], ga[k]), rel(res[0][2][k], res[1][2][k])))
