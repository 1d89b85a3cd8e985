#!/usr/bin/env python
"""forge_attention_fwd vs torch's matmul / softmax / matmul at the 3-D pose estimator's size (B pairs x 4096 tokens x 64 channels), and where the
3-D pose estimator's inference time goes (eager, per top-level piece)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import ops, synthetic as syn  # noqa: E402
from forge_amd.pose_estimator_3d import PoseEstimator3D  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for B in (4, 16):
        q, k, v = (torch.randn(B, 4096, 64, device=dev) * 0.5 for _ in range(3))
        th = timed(lambda: ops.attention(q, k, v))
        tt = timed(lambda: torch.matmul(torch.matmul(q, k.transpose(1, 2)).softmax(dim=-1), v))
        fl = 4.0 * B * 4096 * 4096 * 64
        print("attention B=%d N=4096: forge_attention_fwd %.3f ms (%.1f TF), torch matmul+softmax+matmul %.3f ms" % (B, th, fl / th / 1e9, tt))
    m = PoseEstimator3D(syn.kubric_config())
    m.load_state_dict(syn.seeded_state_dict({"m." + k_: v_ for k_, v_ in m.state_dict().items()}, 0) and {k_[2:]: v_ for k_, v_ in syn.seeded_state_dict({"m." + k_: v_ for k_, v_ in m.state_dict().items()}, 0).items()})
    m = m.to(dev).eval()
    feats = torch.randn(1, 5, 128, 32, 32, 32, device=dev) * 0.5
    print("3-D pose estimator, 1 scene x 5 views: %.3f ms" % timed(lambda: m(feats, return_features=True)))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        m(feats, return_features=True)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=70))

    # ---- the 2-D pose estimator (FPN + 6 attention blocks + stride-2 tail), same question
    from forge_amd.pose_estimator_2d import PoseEstimator2D  # noqa: E402
    m2 = PoseEstimator2D()
    sd2 = syn.seeded_state_dict({"m." + k_: v_ for k_, v_ in m2.state_dict().items()}, 0)
    m2.load_state_dict({k_[2:]: v_ for k_, v_ in sd2.items()})
    m2 = m2.to(dev).eval()
    clips = torch.rand(1, 5, 3, 256, 256, device=dev)
    print("2-D pose estimator, 1 scene x 5 views: %.3f ms; FPN alone %.3f ms" % (timed(lambda: m2(clips, return_features=True)),
                                                                                timed(lambda: m2.backbone.forward_rows(clips[0]))))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        m2(clips, return_features=True)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=26, max_name_column_width=70))
