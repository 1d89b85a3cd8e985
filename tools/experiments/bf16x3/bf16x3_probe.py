#!/usr/bin/env python
"""The Winograd point GEMMs in "f32 via 3 x bf16 split" arithmetic (forge_wino_gemm_bf16x3, convops.bf16x3()) against the exact fp32-MFMA
launches: (1) error of both against a float64 product on the fusion's shapes, (2) time per launch, (3) the configs[1] step - one replay and
PROBE_DEPTH replays in flight - with the switch on and off, and the distance of the rendered views between the two. PROBE_PARTS=err,launch,step"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from forge_amd import convops as co  # noqa: E402
from forge_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
parts = os.environ.get("PROBE_PARTS", "err,launch,step").split(",")
depth = int(os.environ.get("PROBE_DEPTH", "4"))
steps = int(os.environ.get("PROBE_STEPS", "60"))


def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def problem(n, D, Ht, Wt, C1, C2, Cout, kd, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    R = n * D * Ht * Wt
    V1 = torch.randn(16, R, C1, device=dev, generator=g)
    V2 = torch.randn(16, R, C2, device=dev, generator=g) if C2 else None
    U = torch.randn(16, kd, Cout, C1 + C2, device=dev, generator=g) / (kd * (C1 + C2)) ** 0.5
    return V1, V2, U, torch.empty(16, R, Cout, device=dev), torch.empty(16, R, Cout, device=dev)


def run(V1, C1, V2, C2, U, Mm, n, D, Ht, Wt, Cout, x3):
    with co.bf16x3(x3):
        co.wino_gemm(V1, C1, V2, C2, U, Mm, n, D, Ht, Wt, Cout)


if "err" in parts:
    print("== error against float64 (max |err| / max |ref|), kd = 1 (one depth tap: Mm[p] = V[p] U[p]^T) and kd = 3 (fp32 kernel as the yardstick)")
    for (n, D, Ht, Wt, C1, C2, Cout) in ((1, 32, 16, 16, 128, 128, 256), (1, 32, 16, 16, 128, 0, 128), (1, 8, 16, 16, 64, 0, 128), (1, 9, 16, 15, 32, 32, 96)):
        V1, V2, U, M32, M3 = problem(n, D, Ht, Wt, C1, C2, Cout, 1)
        run(V1, C1, V2, C2, U, M32, n, D, Ht, Wt, Cout, False)
        run(V1, C1, V2, C2, U, M3, n, D, Ht, Wt, Cout, True)
        A = (V1 if V2 is None else torch.cat([V1, V2], dim=-1)).double()
        ref = torch.bmm(A, U[:, 0].double().transpose(1, 2))
        sc = ref.abs().max().item()
        e32, e3 = (M32.double() - ref).abs().max().item() / sc, (M3.double() - ref).abs().max().item() / sc
        r32, r3 = (M32.double() - ref).pow(2).mean().sqrt().item() / sc, (M3.double() - ref).pow(2).mean().sqrt().item() / sc
        print("R %6d  C %3d+%3d -> %3d  kd 1: fp32 MFMA max %.3e rms %.3e | bf16x3 max %.3e rms %.3e | tile %s" %
              (n * D * Ht * Wt, C1, C2, Cout, e32, r32, e3, r3, co.wino_gemm_tile(n * D * Ht * Wt, Cout, C1 + C2)), flush=True)
        V1, V2, U, M32, M3 = problem(n, D, Ht, Wt, C1, C2, Cout, 3)
        run(V1, C1, V2, C2, U, M32, n, D, Ht, Wt, Cout, False)
        run(V1, C1, V2, C2, U, M3, n, D, Ht, Wt, Cout, True)
        print("          kd 3: max |bf16x3 - fp32| / max |fp32| = %.3e" % ((M3 - M32).abs().max().item() / M32.abs().max().item()), flush=True)

if "launch" in parts:
    print("== time per launch (HIP events, 20 launches)")
    for name, (n, D, Ht, Wt, C1, C2, Cout) in (("fusion gates  [x|h] -> 256", (1, 32, 16, 16, 128, 128, 256)), ("fusion state  [x|hr] -> 128", (1, 32, 16, 16, 128, 128, 128)),
                                               ("gates h half  128 -> 256", (1, 32, 16, 16, 128, 0, 256)), ("fusion_conv   128 -> 128", (1, 32, 16, 16, 128, 0, 128)),
                                               ("x halves 5 views 128 -> 256", (5, 32, 16, 16, 128, 0, 256)), ("conv1 5 views  64 -> 128", (5, 32, 16, 16, 64, 0, 128)),
                                               ("gates, 4 scenes", (4, 32, 16, 16, 128, 128, 256)), ("gates, 128^3 grid", (1, 64, 32, 32, 128, 128, 256))):
        V1, V2, U, M32, M3 = problem(n, D, Ht, Wt, C1, C2, Cout, 3)
        gf = 2.0 * 16 * n * D * Ht * Wt * Cout * 3 * (C1 + C2) / 1e9
        t32 = timed(lambda: run(V1, C1, V2, C2, U, M32, n, D, Ht, Wt, Cout, False), 20)
        t3 = timed(lambda: run(V1, C1, V2, C2, U, M3, n, D, Ht, Wt, Cout, True), 20)
        print("%-30s %7.1f GF  fp32 %.3f ms (%5.1f TF)  bf16x3 %.3f ms (%5.1f TF-equivalent)  x%.2f" % (name, gf, t32, gf / t32, t3, gf / t3, t32 / t3), flush=True)
        del V1, V2, U, M32, M3
        torch.cuda.empty_cache()

if "step" in parts:
    from forge_amd.graph import GraphedForward, PipelinedForward
    from forge_amd.model import FORGE
    print("== configs[1] step (1 scene, 5 views in / 5 out)")
    cfg, ds = syn.kubric_config(), syn.SyntheticDataset(1.5)
    sample = {k: v.to(dev) for k, v in syn.make_sample(1, 5, 256, 1.5, seed=1000).items()}
    m = FORGE(cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()

    def wall(fn, n, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    outs = {}
    for x3 in (False, True, False):
        with co.bf16x3(x3):
            g = GraphedForward(m, sample, ds, dev)
            outs[x3] = [o.clone() for o in g(sample)[:2]]
            one = min(wall(lambda: g(sample), steps, 5) for _ in range(3))
            del g
            torch.cuda.empty_cache()
            p = PipelinedForward(m, sample, ds, dev, depth=depth, warmup=1)
            pipe = min(wall(lambda: p(sample), 2 * steps, depth) for _ in range(3))
            del p
            torch.cuda.empty_cache()
        print("bf16x3 %-5s one replay %7.3f ms | %d in flight %7.3f ms/step = %7.1f views/s" % (x3, one, depth, pipe, 5e3 / pipe), flush=True)
    d = (outs[True][0] - outs[False][0]).abs().max().item()
    mse = (outs[True][0] - outs[False][0]).pow(2).mean().item()
    print("rendered rgb: max |bf16x3 - fp32| %.3e, PSNR between the two %.1f dB" % (d, 10 * torch.log10(torch.tensor(1.0 / max(mse, 1e-30))).item()))

if "replay" in parts:                      # for rocprofv3 --kernel-trace --stats: PROBE_X3=0|1, 30 one-stream replays of the configs[1] step
    from forge_amd.graph import GraphedForward
    from forge_amd.model import FORGE
    cfg, ds = syn.kubric_config(), syn.SyntheticDataset(1.5)
    sample = {k: v.to(dev) for k, v in syn.make_sample(1, 5, 256, 1.5, seed=1000).items()}
    m = FORGE(cfg)
    m.load_state_dict(syn.seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()
    with co.bf16x3(os.environ.get("PROBE_X3", "1") == "1"):
        g = GraphedForward(m, sample, ds, dev)
        for _ in range(30):
            g(sample)
        torch.cuda.synchronize()
