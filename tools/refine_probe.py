#!/usr/bin/env python
"""Time the pose-refinement loop (kubric_eval.py do_refinement shape: 1 scene, 5 views, 4 free poses) on one MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import geo_utils, refine, synthetic as syn  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

dev = torch.device("cuda:0")
t = int(os.environ.get("REFINE_VIEWS", "5"))
iters = int(os.environ.get("REFINE_ITERS", "30"))
cfg = syn.kubric_config()
model = FORGE(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).eval()
ds = syn.SyntheticDataset(1.5)
sample = syn.make_sample(1, t, 256, 1.5, seed=21)
with torch.no_grad():
    feats = model.encoder_3d.get_feat3D(sample["images"][0].to(dev)).reshape(1, t, 128, 32, 32, 32)
    gt7 = geo_utils.mat2quat(sample["cam_poses_rel_cv2"][0, 1:]).to(dev)
    tgt_i, tgt_m, _, _, _ = refine._render_views(model, cfg, ds, feats, gt7, sample["K_cv2"].to(dev), dev)
init = gt7.clone()
init[:, :4] = torch.nn.functional.normalize(init[:, :4] + 0.03 * torch.randn(t - 1, 4, device=dev))
init[:, 4:] += 0.02 * torch.randn(t - 1, 3, device=dev)
e0 = refine.pose_errors(init, sample["cam_poses_rel_cv2"][0, 1:].to(dev))
use_graph = os.environ.get("REFINE_GRAPH", "1") != "0"
out, hist, dt = refine.refine_poses(model, cfg, ds, feats, init, tgt_i, tgt_m, sample["K_cv2"], dev, iter_num=iters, log_every=10, use_graph=use_graph)
e1 = refine.pose_errors(out, sample["cam_poses_rel_cv2"][0, 1:].to(dev))
print("refinement t=%d (%s): %.1f ms/iteration; loss %.5f -> %.5f; rot err %.2f -> %.2f deg; trans err %.4f -> %.4f"
      % (t, "hipGraph replay" if use_graph else "eager", dt * 1e3, hist[0], hist[-1], e0[0].mean().item(), e1[0].mean().item(), e0[1].mean().item(), e1[1].mean().item()))
