"""Where do the small torch ops of the training step come from? A TorchDispatchMode logs every aten op of ONE step (forward, loss, backward, clip, Adam)
with the innermost forge_amd / tools / torch.optim / clip_grad frame of the Python stack at dispatch time; autograd-engine ops (gradient accumulation,
backward of torch ops) show up without a Python frame."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from forge_amd import synthetic as syn
from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
from forge_amd.train import grouped_mse

b = int(os.environ.get("TRAIN_SCENES", "1"))
dev = torch.device("cuda:0")
cfg = syn.kubric_config()
model = FORGE_poseEstimator3D(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=3).items()}
ds = syn.SyntheticDataset(1.5)
WANT = ("copy_", "add_", "fill_", "cat", "add.", "zeros", "zero_", "clone", "contiguous", "sum", "mul", "threshold", "clamp_min", "relu", "empty_like", "index", "stack")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(w in name for w in WANT):
            fr = "(no python frame: autograd engine)"
            for f in reversed(traceback.extract_stack(limit=40)):
                fn = f.filename
                if ("forge_amd" in fn or "torch/optim" in fn or "clip_grad" in fn) and "train_dispatch_trace" not in fn:
                    fr = "%s:%d %s" % (os.path.relpath(fn, ROOT) if fn.startswith(ROOT) else fn[-40:], f.lineno, f.name)
                    break
            numel = max([a.numel() for a in args if torch.is_tensor(a)] + [0])
            self.agg[(name, fr, "big" if numel > 1 << 16 else "small")] += 1
        return func(*args, **(kwargs or {}))


def step():
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
log = Log()
with log:
    step()
torch.cuda.synchronize()
print("ops logged:", sum(log.agg.values()))
for (name, fr, size), n in log.agg.most_common(70):
    print("%5d  %-32s %-5s %s" % (n, name[:32], size, fr))
