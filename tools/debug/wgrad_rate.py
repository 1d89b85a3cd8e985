"""TF of the weight-gradient launches of the ConvGRU at 1 and 4 scenes: direct (conv_wgrad, K = voxels, 27 taps) and Winograd (conv3_wgrad:
transform of x / h, wino_dy, 16 batched point problems of conv_wgrad_kernel, G^T dU G) - executed FLOPs / time of the whole call and of the
wgrad kernel alone (HIP events around the call; the transforms are reported by rocprofv3 runs of tools/train_step_probe.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from forge_amd import convops as co
dev = torch.device("cuda:0")


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


D, C = 32, 128
for n in (1, 4):
    M = n * D ** 3
    x, h = torch.randn(n, D, D, D, C, device=dev), torch.randn(n, D, D, D, C, device=dev)
    for Cout, name in ((256, "gates"), (128, "candidate")):
        dy = torch.randn(n, D, D, D, Cout, device=dev)
        dwp = torch.zeros(27, Cout, 2 * C, device=dev)
        fl = 2.0 * M * Cout * 2 * C * 27
        ms_d = timed(lambda: co.conv_wgrad(dy, x, C, h, C, dwp, (n, D, D, D), (D, D, D), Cout, co.TAPS_3x3x3))
        ms_w = timed(lambda: co.conv3_wgrad(dy, x, C, h, C, dwp, (n, D, D, D), Cout))
        print("scenes %d %-9s direct wgrad %.3f ms %.1f TF | Winograd wgrad (all five launches) %.3f ms = %.1f TF executed (%.1f TF direct-equivalent)"
              % (n, name, ms_d, fl / ms_d / 1e9, ms_w, fl / 2.25 / ms_w / 1e9, fl / ms_w / 1e9))
