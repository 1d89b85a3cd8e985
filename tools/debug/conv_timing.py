"""Where does a short conv_igemm launch spend its time? Runs ResNet-trunk-sized GEMMs on the instrumented library
(tools/debug/build_timing_lib.sh) and prints, per shape: event-timed launch duration and, from the per-workgroup clock stamps
(wall_clock64, 100 MHz), the span first-entry -> last-exit and the median per-workgroup phases prologue / main loop / epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FORGE_AMD_LIB"] = os.path.join(ROOT, "tools", "debug", "libforge_hip_timing.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from forge_amd import _lib, convops as co
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); from _variants import apply_environ  # noqa: E402,E702
dev = torch.device("cuda:0")
L = _lib.lib()
L.forge_debug_conv_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
shapes = [(5120, 512, 128, 1), (5120, 128, 512, 1), (5120, 128, 128, 9), (5120, 1024, 256, 1), (5120, 256, 256, 9), (5120, 512, 512, 9), (20480, 64, 64, 1),
          (32768, 128, 256, 27),
          # the per-workgroup shape of the Winograd gates launch (16 points x 8192 rows as ONE 3-depth-tap problem, K = 768, 24 K-steps), forced 64x64 tile
          (131072, 256, 256, 3)]
for M, N, K, T in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(T, N, K, device=dev) * 0.02
    sc, sh = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    taps = [(0, 0, 0)] if T == 1 else ([(0, dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)] if T == 9 else co.TAPS_3x3x3)
    side = int(round((M / 5) ** 0.5)) if T != 27 else 32
    grid = (5, 1, side, side) if T != 27 else (1, 32, 32, 32)
    os.environ["FORGE_CONV_KSPLIT"] = "1"
    apply_environ()
    os.environ.pop("FORGE_CONV_TILE", None)
    apply_environ()
    if T == 3:
        taps, grid = [(-1, 0, 0), (0, 0, 0), (1, 0, 0)], (16, 32, 16, 16)
        os.environ["FORGE_CONV_TILE"] = "D"
        apply_environ()
    epi = co.EPI_BIAS if T == 3 else co.EPI_AFFINE_ACT          # the point GEMMs store raw products
    f = lambda: co.conv_igemm(x, K, K, None, 0, 0, w, None, sc, sh, 0.0, None, None, None, out, None, grid, grid[1:], N, N, taps, epilogue=epi)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    tile, ks = co.conv_plan(M, N, K, T, co.EPI_AFFINE_ACT, N)
    tile, ks = (os.environ.get("FORGE_CONV_TILE") or tile), 1
    bm, bn = {"A": (128, 128), "B": (64, 128), "C": (128, 64), "D": (64, 64), "E": (128, 32), "F": (128, 64)}[tile]
    nwg = -(-M // bm) * -(-N // bn)
    buf = np.zeros((min(nwg, 8192), 4), dtype=np.int64)
    L.forge_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0])
    tick = 0.01                                           # us per wall_clock64 tick (100 MHz)
    span = (buf[:, 3].max() - buf[:, 0].min()) * tick
    pro, loop, epi = [np.median(buf[:, i + 1] - buf[:, i]) * tick for i in range(3)]
    start_spread = (buf[:, 0].max() - buf[:, 0].min()) * tick
    print("M=%-6d N=%-5d K=%-5d tile %s x%d wgs=%-5d event %.1f us | first-entry..last-exit %.1f us, entry spread %.1f | per-WG median: prologue %.1f  loop %.1f  epilogue %.1f us"
          % (M, N, K * T, tile, ks, nwg, a.elapsed_time(b) * 1e3, span, start_spread, pro, loop, epi))
