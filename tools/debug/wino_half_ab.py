"""A/B in one process: the configs[1] step with the inverse transform's row stage in the GEMM epilogue (forge_wino_gemm_half / forge_wino_output_half:
8 point-product planes through HBM) against the two-launch form with 16 planes (convops.wino_half_applies forced to False). One replay and
PROBE_DEPTH replays in flight, outputs compared bit for bit. PROBE_MODEL=forge|pose3d"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.graph import GraphedForward, PipelinedForward  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: E402

dev = torch.device("cuda:0")
steps, depth = int(os.environ.get("PROBE_STEPS", "100")), int(os.environ.get("PROBE_DEPTH", "4"))
scenes = int(os.environ.get("PROBE_SCENES", "1"))
cls = FORGE_poseEstimator3D if os.environ.get("PROBE_MODEL", "forge") == "pose3d" else FORGE
cfg, ds = syn.kubric_config(), syn.SyntheticDataset(1.5)
sample = {k: v.to(dev) for k, v in syn.make_sample(scenes, 5, 256, 1.5, seed=1000).items()}
m = cls(cfg)
m.load_state_dict(syn.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
applies = co.wino_half_applies


def wall(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = None
for label, fn in (("16 planes (two-launch form)", lambda *a: False), ("8 planes (row stage in the GEMM)", applies)) * 2:
    co.wino_half_applies = fn
    g = GraphedForward(m, sample, ds, dev)
    out = [o.clone() for o in g(sample)[:2]]
    same = "" if ref is None else ("bit-identical" if all(torch.equal(a, b) for a, b in zip(out, ref)) else "DIFFERENT")
    ref = ref or out
    one = min(wall(lambda: g(sample), steps, 5) for _ in range(3))
    del g
    torch.cuda.empty_cache()
    p_ = PipelinedForward(m, sample, ds, dev, depth=depth, warmup=1)
    pipe = min(wall(lambda: p_(sample), 2 * steps, depth) for _ in range(3))
    del p_
    torch.cuda.empty_cache()
    print("%-34s one replay %7.3f ms | %d in flight %7.3f ms/step = %7.1f views/s %s" % (label, one, depth, pipe, 5e3 * scenes / pipe, same), flush=True)
co.wino_half_applies = applies
