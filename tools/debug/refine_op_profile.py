"""torch.profiler view of ONE eager pose-refinement iteration: which aten ops surround the hand-written kernels (pose algebra, view ordering, camera
packing, losses, Adam) and how many launches they cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from forge_amd import geo_utils, refine, synthetic as syn
from forge_amd.model import FORGE

dev = torch.device("cuda:0")
t = 5
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
frozen = refine._frozen(model)
r = refine.PoseRefiner(model, cfg, ds, feats, gt7 + 0.01, tgt_i, tgt_m, sample["K_cv2"], dev, use_graph=True)
for _ in range(3):
    r.eager_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    r.eager_step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("device time of one iteration: %.2f ms; launches by op:" % (tot / 1e3))
for e in rows[:60]:
    print("%-70s %5d calls %9.1f us %5.1f%%" % (e.key[:70], e.count, e.self_device_time_total, 100 * e.self_device_time_total / tot))
