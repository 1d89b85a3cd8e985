#!/usr/bin/env python
"""A/B of the ConvGRU fusion (Encoder3D.fuse, 5 views, 128 channels) on the Winograd F(2x2,3x3) x 3-depth-tap path vs the direct
implicit-GEMM path, hipGraph replay; then per-launch HIP-event times of one eager Winograd pass. WINO_SCENES / WINO_GRID (32 | 64)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import _lib, convops as co, synthetic as syn  # noqa: E402
from forge_amd.fusion import ConvGRU_3D  # noqa: E402
from forge_amd.graph import GraphedCall  # noqa: E402

b = int(os.environ.get("WINO_SCENES", "1"))
D = int(os.environ.get("WINO_GRID", "32"))
t, C = 5, 128
dev = torch.device("cuda:0")
gru = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=C, hidden_size=C)
gru.load_state_dict(syn.seeded_state_dict(gru.state_dict(), 1))
gru = gru.to(dev).eval()
x = (torch.randn(b, t, D, D, D, C, device=dev) * 0.5).permute(0, 1, 5, 2, 3, 4)
flops = 2.0 * b * D ** 3 * 27 * C * C * (2 + t * (4 + 2))          # direct-convolution FLOPs of the fusion (fc0, fc3, t x (gates, state))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


outs = {}
for mode in ("0", "1"):
    with torch.no_grad(), co.winograd(mode == "1"):
        g = GraphedCall(lambda: gru.fuse_hip(x), dev, warmup=2)
        outs[mode] = g().clone()
        ms = timed(g)
    print("scenes %d grid %d^3 fusion %-8s %8.3f ms   %.1f direct-equivalent TF" % (b, D, "winograd" if mode == "1" else "direct", ms, flops / ms / 1e9))
print("max |winograd - direct| = %.3e (output max %.3f)" % ((outs["1"] - outs["0"]).abs().max().item(), outs["0"].abs().max().item()))

# per-launch times of the Winograd pieces (eager, HIP events around each launch)
rec = []


def wrap(name):
    fn = getattr(co, name)

    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        if name == "wino_gemm":
            tag = "gemm N=%d K=%d" % (a[10], 3 * (a[1] + a[3]))
        elif name == "wino_input":
            tag = "input n=%d" % a[3]
        else:
            tag = "output N=%d epi=%d" % (a[15], a[17])
        rec.append((tag, e0, e1))
        return r
    setattr(co, name, w)


for nm in ("wino_input", "wino_gemm", "wino_output"):
    wrap(nm)
with torch.no_grad():
    gru.fuse_hip(x)
    rec.clear()
    gru.fuse_hip(x)
torch.cuda.synchronize()
agg = {}
for tag, e0, e1 in rec:
    a = agg.setdefault(tag, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
R = b * D * (D // 2) * (D // 2)
for tag, (n, ms) in agg.items():
    extra = ""
    if tag.startswith("gemm"):
        N, K = int(tag.split("N=")[1].split()[0]), int(tag.split("K=")[1])
        extra = "  %.1f TF (MFMA FLOPs)" % (2.0 * 16 * R * N * K * n / ms / 1e9)
    print("  %-24s x%-2d %8.3f ms total  %7.3f ms each%s" % (tag, n, ms, ms / n, extra))
