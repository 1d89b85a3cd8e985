"""TIMING-ONLY emulation of a full 3-D Winograd F(2,3)^3 fusion inside the configs[1] step (results are wrong, data stays finite): what would the
step - one replay and 4 replays in flight - cost if the ConvGRU / fusion_conv convolutions ran as 64 one-tap point GEMMs over half as many tile rows
(1.5x fewer multiplies than the committed F(2x2,3x3) x 3 depth taps) with 8x instead of 4x operand expansion? Emulated with the existing kernels at
equal work and bytes per convolution: input transform twice (64 x R/2 x C = 2 x 16 x R x C floats written), the 16-point GEMM launch twice with ONE
depth tap over all R rows (= 64 points x R/2 rows, K = Cin), inverse transform twice (2 x 16 x R x Cout floats read). tools/debug/wino3d_feasibility.py
timed the GEMMs alone (round 2); this puts the whole launch sequence into the pipelined step, where HBM-bound transforms hide behind other steps' GEMMs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.fusion import ConvGRU_3D  # noqa: E402
from forge_amd.graph import GraphedForward, PipelinedForward  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

dev = torch.device("cuda:0")
steps, depth = int(os.environ.get("PROBE_STEPS", "60")), int(os.environ.get("PROBE_DEPTH", "4"))
TWICE_IN, TWICE_OUT = int(os.environ.get("EMU_IN", "2")), int(os.environ.get("EMU_OUT", "2"))     # 1 = keep the 2-D form's transform bytes (GEMM change alone)


def emu_h0(p, src, geo, Vh, Mc, t0, h, nsum=1, sum_stride=0, bs=0):
    b, D, H, W = geo
    C = h.shape[-1]
    U0, U3 = p["fc0_U1"], p["fc3_U1"]
    for _ in range(TWICE_IN):
        co.wino_input(src, C, C, b, D, H, W, bs=bs, out=Vh, nsum=nsum, sum_stride=sum_stride)
    for _ in range(2):
        co.wino_gemm(Vh, C, None, 0, U0, Mc, b, D, H // 2, W // 2, C)
    for _ in range(TWICE_OUT):
        co.wino_output(Mc, p["fc0_b"], p["bn1"][0], p["bn1"][1], 0.01, None, None, None, t0, None, None, b, D, H, W, C, C, co.EPI_AFFINE_ACT)
    for _ in range(TWICE_IN):
        co.wino_input(t0, C, C, b, D, H, W, out=Vh)
    for _ in range(2):
        co.wino_gemm(Vh, C, None, 0, U3, Mc, b, D, H // 2, W // 2, C)
    for _ in range(TWICE_OUT):
        co.wino_output(Mc, p["fc3_b"], p["bn4"][0], p["bn4"][1], 0.01, None, None, None, h, None, None, b, D, H, W, C, C, co.EPI_AFFINE_ACT)


def emu_fuse(self, xr, h0=None):
    b, t, D, H, W, C = xr.shape
    p = self._packed_wino()
    if "gate_U1" not in p:                                             # one depth tap of each transformed weight: K = Cin per point
        for k in ("gate", "out", "fc0", "fc3"):
            p[k + "_U1"] = p[k + "_U"][:, 1:2].contiguous()
    dev_, M, Ht, Wt = xr.device, b * D * H * W, H // 2, W // 2
    R = b * D * Ht * Wt
    new = lambda c=C: torch.zeros(M, c, dtype=torch.float32, device=dev_)
    geo = (b, D, H, W)
    Vx = None
    for _ in range(TWICE_IN):
        Vx = co.wino_input(xr, C, C, b * t, D, H, W)
    Vh = torch.zeros(16, R, C, dtype=torch.float32, device=dev_)
    Mm = torch.zeros(16, R, 2 * C, dtype=torch.float32, device=dev_)
    Mc = Mm.view(-1)[:16 * R * C].view(16, R, C)
    t0, h = new(), new()
    vol = D * H * W
    emu_h0(p, xr, geo, Vh, Mc, t0, h, nsum=t, sum_stride=vol, bs=t * vol)
    z, hr, h2, out = new(), new(), t0, new()
    for ti in range(t):
        for _ in range(TWICE_IN):
            co.wino_input(h, C, C, b, D, H, W, out=Vh)
        for _ in range(2):
            co.wino_gemm(Vx, C, Vh, C, p["gate_U1"], Mm, b, D, Ht, Wt, 2 * C, view=ti, views=t)
        for _ in range(TWICE_OUT):
            co.wino_output(Mm, p["gate_b"], None, None, 1.0, None, h, None, z, hr, None, *geo, 2 * C, C, co.EPI_GRU_GATES)
        for _ in range(TWICE_IN):
            co.wino_input(hr, C, C, b, D, H, W, out=Vh)
        for _ in range(2):
            co.wino_gemm(Vx, C, Vh, C, p["out_U1"], Mc, b, D, Ht, Wt, C, view=ti, views=t)
        last = ti == t - 1
        for _ in range(TWICE_OUT):
            co.wino_output(Mc, p["out_b"], p["norm"][0], p["norm"][1], 1.0, None, h, z, h2, out if last else None, None, *geo, C, C, co.EPI_GRU_OUT)
        h, h2 = h2, h
    return out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3)


def wall(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


cfg, ds = syn.kubric_config(), syn.SyntheticDataset(1.5)
sample = {k: v.to(dev) for k, v in syn.make_sample(1, 5, 256, 1.5, seed=1000).items()}
m = FORGE(cfg)
m.load_state_dict(syn.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
orig = ConvGRU_3D._fuse_wino
for label, fn in (("committed 2-D x 3 depth taps", orig), ("3-D form emulated (timing only)", emu_fuse), ("committed 2-D x 3 depth taps", orig)):
    ConvGRU_3D._fuse_wino = fn
    g = GraphedForward(m, sample, ds, dev)
    finite = bool(torch.isfinite(g(sample)[0]).all())
    one = min(wall(lambda: g(sample), steps, 5) for _ in range(3))
    del g
    torch.cuda.empty_cache()
    p_ = PipelinedForward(m, sample, ds, dev, depth=depth, warmup=1)
    pipe = min(wall(lambda: p_(sample), 2 * steps, depth) for _ in range(3))
    del p_
    torch.cuda.empty_cache()
    print("%-34s one replay %7.3f ms | %d in flight %7.3f ms/step = %7.1f views/s | output finite: %s" % (label, one, depth, pipe, 5e3 / pipe, finite), flush=True)
ConvGRU_3D._fuse_wino = orig
