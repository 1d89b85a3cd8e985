#!/usr/bin/env python
"""How much would running INDEPENDENT fusion chains of the training step on separate HIP streams buy (their HBM-bound Winograd transforms / GRU
element-wise kernels under another chain's MFMA-bound point GEMMs)? Two ConvGRU fusions (forward + backward, train-mode BatchNorm) of PROBE_SCENES
scenes x 5 views each: back to back on one stream vs one per stream from two Python threads. Upper bound for a multi-stream `_FuseGroupsTrain`."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.fusion import ConvGRU_3D  # noqa: E402

dev = torch.device("cuda:0")
b = int(os.environ.get("PROBE_SCENES", "4"))
torch.manual_seed(0)
gru = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=128, hidden_size=128).to(dev).train()
xs = [(torch.randn(b, 5, 128, 32, 32, 32, device=dev) * 0.5).requires_grad_(True) for _ in range(2)]


def work(x, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        y = gru.fuse_autograd_hip(x)
        y.square().sum().backward()


def sequential():
    for x in xs:
        work(x)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def concurrent():
    ths = [threading.Thread(target=work, args=(x, s)) for x, s in zip(xs, streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


def timed(fn, n=5):
    for _ in range(2):
        fn()
        for p in gru.parameters():
            p.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        for p in gru.parameters():
            p.grad = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = timed(sequential)
c = timed(concurrent)
print("two 5-view ConvGRU fusions, forward + backward, %d scene(s) each: one stream %.2f ms, two streams / threads %.2f ms (%.1f %%)" % (b, a, c, 100 * (c / a - 1)))
