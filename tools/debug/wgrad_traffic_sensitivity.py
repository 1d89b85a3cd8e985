#!/usr/bin/env python
"""Is the Winograd weight-gradient launch (forge_wino_wgrad -> conv_wgrad_kernel<128, 1>) held back by its operand re-fetch (VERDICT r4 item 3:
~2.2-3x its algorithmic bytes through the L2 -> fabric port)? Timing-only experiment, results meaningless: the same launch with the 16
Winograd points' V operands ALIASED onto one point (point stride 1 float instead of R x Cin: the V side of the traffic becomes L2 / MALL
resident, 1/16 of the bytes), next to the real launch, at the ConvGRU gates shape of the 4-scene training step and at one scene."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L, p = _lib.lib(), _lib.ptr


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for n, label in ((2, "gates wgrad, M = 16384 tile rows (the 10 x 853 us launches of the 4-scene step)"), (4, "the same at 4 scenes' rows"), (1, "one scene")):
    D, Ht, Wt, C, Cout = 32, 16, 16, 256, 256
    R = n * D * Ht * Wt
    dM = torch.randn(16, R, Cout, device=dev)
    V = torch.randn(16, R, C, device=dev)
    dU = torch.zeros(16, 3, Cout, C, device=dev)
    st = _lib.current_stream()
    real = timed(lambda: _lib.check(L.forge_wino_wgrad(p(dM), p(V), C, 0, 0, None, 0, 0, 0, p(dU), n, D, Ht, Wt, Cout, 3, st), "w"))
    alias = timed(lambda: _lib.check(L.forge_wino_wgrad(p(dM), p(V), C, 0, 4, None, 0, 0, 0, p(dU), n, D, Ht, Wt, Cout, 3, st), "w"))
    fl = 2.0 * 16 * 3 * R * Cout * C
    print("%-78s real %.1f us (%.1f TF)   V aliased over the points %.1f us (%.1f TF)   algorithmic %.0f MB" %
          (label, real * 1e3, fl / real / 1e9, alias * 1e3, fl / alias / 1e9, (dM.numel() + V.numel()) * 4 / 1e6))
