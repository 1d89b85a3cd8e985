"""Per-kernel measurements of the headline step: HIP-event recorders around the stages / conv launches, the hand-written kernels
at the bench shapes (algorithmic bytes or FLOPs per launch / average duration), and readers of the committed rocprofv3 evidence."""
import json
import os

import torch

from forge_amd import _lib, synthetic as syn
from benchkit.common import FP32_MFMA_PEAK_TF, HBM_PEAK_GBS, ROOT, T_IN, V_OUT, time_kernel


def stage_timers(model):
    """HIP events around the hot-path stages, recorded on the current (launch) stream. Wraps the sub-module
    entry points FORGE.forward calls; returns (records, undo)."""
    rec, undo = {}, []

    def wrap(obj, attr, name):
        fn = getattr(obj, attr)

        def timed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            rec.setdefault(name, []).append((e0, e1))
            return out
        setattr(obj, attr, timed)          # instance attribute shadows the class method
        undo.append(lambda: delattr(obj, attr))

    e3 = model.encoder_3d
    wrap(e3, "_trunk_hip", "encoder_resnet")
    wrap(e3, "get_feat3D", "encoder_total")
    wrap(model.rotate, "forward", "rotate")
    wrap(e3, "fuse", "fuse")
    wrap(e3, "heads", "heads")
    wrap(model.render, "forward", "render_total")
    wrap(model.render, "_conv_rgb_hip", "conv_rgb")
    # every forge_conv_igemm launch: events + algorithmic FLOPs, keyed by kernel instantiation
    from forge_amd import convops as co
    orig = co.conv_igemm

    def conv_timed(in1, C1, ld1, in2, C2, ld2, wp, *a, **kw):
        grid, Cout, taps = a[9], a[11], a[13]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(in1, C1, ld1, in2, C2, ld2, wp, *a, **kw)
        e1.record()
        M = grid[0] * grid[1] * grid[2] * grid[3]
        # the plan forge_conv_igemm itself uses (forge_conv_igemm_plan): names match the rocprofv3 kernel names; a split-K launch
        # (GEMM + reduction kernel) is attributed to its GEMM instantiation
        nphase = 1
        if tuple(kw.get("phase", (0, 0, 0))) == (-1, -1, -1):       # merged transposed-conv phases: 8 (3-D) or 4 (2-D, D not doubled)
            nphase = 8 if kw["out_grid"][0] == 2 * grid[1] else 4
        tile, ksplit = co.conv_plan(M, Cout, C1 + C2, len(taps), kw.get("epilogue", co.EPI_BIAS), a[12], nphase)
        key = "conv_igemm_n16_kernel + conv_igemm_n16_lines_kernel<R> (Cout <= 16)" if tile == "N" else "conv_igemm_kernel<%s>" % co.TILE_NAMES[tile]
        rec.setdefault(key, []).append((e0, e1, 2.0 * M * Cout * len(taps) * (C1 + C2), (M, Cout, len(taps), C1 + C2)))
        return out
    co.conv_igemm = conv_timed
    undo.append(lambda: setattr(co, "conv_igemm", orig))
    # the Winograd path of the ConvGRU fusion: its 16 point GEMMs are ONE launch of the same conv_igemm_kernel (counted above with the
    # MFMA FLOPs they execute, 2 x 16 R x Cout x 3 Cin - 2.25x fewer than the direct convolution they replace); the two transform
    # kernels are HBM-bound and recorded with their algorithmic bytes
    o_g, o_i, o_o = co.wino_gemm, co.wino_input, co.wino_output

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def gemm_timed(V1, C1, V2, C2, U, Mm, n, D, Ht, Wt, Cout, **kw):
        e0, e1 = ev()
        e0.record()
        out = o_g(V1, C1, V2, C2, U, Mm, n, D, Ht, Wt, Cout, **kw)
        e1.record()
        R = n * D * Ht * Wt
        rec.setdefault("conv_igemm_kernel<%s>" % co.TILE_NAMES[co.wino_gemm_tile(R, Cout, C1 + C2)], []).append((e0, e1,
                2.0 * 16 * R * Cout * U.shape[1] * (C1 + C2), (16 * R, Cout, U.shape[1], C1 + C2), 2.25))
        return out

    def input_timed(x, C, ld, n, D, H, W, **kw):
        e0, e1 = ev()
        e0.record()
        out = o_i(x, C, ld, n, D, H, W, **kw)
        e1.record()
        # reads the rows (of nsum views) once, writes 16 points x R = 4x
        rec.setdefault("wino_input_kernel", []).append((e0, e1, 4.0 * n * D * H * W * C * (kw.get("nsum", 1) + 4)))
        return out

    def output_timed(Mm, bias, scale, shift, slope, residual, aux_h, aux_z, out, out2, out3, n, D, H, W, Cout, ldo, epilogue, **kw):
        e0, e1 = ev()
        e0.record()
        r = o_o(Mm, bias, scale, shift, slope, residual, aux_h, aux_z, out, out2, out3, n, D, H, W, Cout, ldo, epilogue, **kw)
        e1.record()
        rows = n * D * H * W
        side = {co.EPI_GRU_GATES: (2 if out2 is not None else 1) * Cout // 2 + Cout // 2, co.EPI_GRU_OUT: 3 * Cout + (Cout if out2 is not None else 0)}.get(
            epilogue, Cout if out is not None else 0)
        side += 4 * Cout if kw.get("Mm2") is not None else 0
        # reads 16 points x R x Cout = 4x (8 planes = 2x when the row stage ran in the GEMM epilogue), then the tail's operands
        rec.setdefault("wino_output_kernel", []).append((e0, e1, 4.0 * rows * ((2 if kw.get("half") else 4) * Cout + side)))
        return r
    co.wino_gemm, co.wino_input, co.wino_output = gemm_timed, input_timed, output_timed
    undo.append(lambda: (setattr(co, "wino_gemm", o_g), setattr(co, "wino_input", o_i), setattr(co, "wino_output", o_o)))
    return rec, undo


def pmc_traffic(prefix):
    """HBM bytes per launch of the kernel whose summary key contains `prefix`, from the committed rocprofv3 PMC passes (profiles/*pmc_summary.json: separate
    --pmc
    FETCH_SIZE / WRITE_SIZE runs of tools/probe_kernels.py, FETCH_SIZE doubled per MI355X_MICROARCH.md). PMC counters
    cannot be read from inside this process; null when no summary is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if prefix in k and isinstance(v, dict) and "hbm_bytes_corrected" in v:
                return {"hbm_bytes_per_launch": v["hbm_bytes_corrected"], "algorithmic_bytes": v.get("algorithmic_bytes"),
                        "launch": k, "source": os.path.basename(f)}
    return None


def rocprof_conv_time():
    """Per-step kernel time of the dominant kernel from the committed `rocprofv3 --kernel-trace --stats` run of this command with ONE step in flight
    (profiles/r*_rocprofv3_kernel_stats.csv + its .meta.json: steps traced): sum of TotalDurationNs over every conv_igemm_kernel<...> instantiation /
    steps - pure kernel durations (no launch gaps), what the eager HIP-event pairs of `frac` cannot give. None when no profile is committed."""
    import csv
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats.csv")), reverse=True):
        meta = f[:-4] + ".meta.json"
        if "grid64" in f or "in_flight" in f or not os.path.exists(meta):
            continue
        try:
            m = json.load(open(meta))
            rows = [r for r in csv.DictReader(open(f)) if "conv_igemm_kernel<" in r["Name"]]
            ns = sum(float(r["TotalDurationNs"]) for r in rows)
            return {"ms_per_step": ns / 1e6 / m["steps_traced"], "launches_per_step": sum(int(r["Calls"]) for r in rows) / m["steps_traced"],
                    "source": os.path.basename(f), "steps_traced": m["steps_traced"]}
        except Exception:
            continue
    return None


def kernel_rooflines(dev, B, D=32):
    """Hand-written kernels at the bench shapes: ALGORITHMIC bytes per launch / avg duration. D = feature grid (32 -> 64^3 render
    volume, 64 -> 128^3). The HBM-bound kernels cycle through NBUF distinct source/destination sets whose total exceeds the 256 MB
    Infinity Cache, so `ms` is an HBM number as inside the real step (a re-launch on one 168 MB set is served from the MALL:
    26.7 us vs 42 us in the step)."""
    lib = _lib.lib()
    st = _lib.current_stream()
    out = {}
    # rotate: n = B*5 volumes of [D^3, 128]; 4 warped (read + write) + 1 copied per scene
    C, n = 128, B * T_IN
    set_bytes = n * C * D ** 3 * 4 * 2
    nbuf = max(2, min(8, -(-(768 << 20) // set_bytes)))
    srcs = [torch.randn(n, D, D, D, C, device=dev) for _ in range(nbuf)]
    dsts = [torch.empty_like(srcs[0]) for _ in range(nbuf)]
    xf = torch.tensor([1, 0, 0, 0.02, 0, 0.8, -0.6, 0, 0, 0.6, 0.8, 0.01], device=dev).repeat(n, 1).contiguous()
    mode = torch.ones(n, dtype=torch.int32, device=dev)
    mode[::T_IN] = 0
    it = [0]

    def rot():
        k = it[0] % nbuf
        it[0] += 1
        _lib.check(lib.forge_rotate_fwd(_lib.ptr(srcs[k]), _lib.ptr(xf), _lib.ptr(mode), _lib.ptr(dsts[k]), n, C, D, D, D, st), "rotate")
    ms = time_kernel(rot, iters=4 * nbuf, warm=nbuf)
    out["rotate_fwd_kernel"] = {"bound": "hbm", "ms": ms, "bytes": set_bytes, "achieved": set_bytes / ms / 1e6, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": set_bytes / ms / 1e6 / HBM_PEAK_GBS, "working_set_mb": nbuf * set_bytes / 2 ** 20,
                                "traffic": pmc_traffic("rotate_fwd_kernel")}
    del srcs, dsts
    # render: B volumes (2D)^3 x (16+1), V = 5 views each, 128^2 rays, 64 samples
    Dr, Cr, V = 2 * D, 16, B * V_OUT
    vol_bytes = B * 17 * Dr ** 3 * 4
    nbuf = max(1, min(8, -(-(512 << 20) // vol_bytes))) if D > 32 else 1        # 64^3: the volume was just written by the heads (MALL-warm in the step too)
    feat0, dens0 = syn.blob_volumes(B, Dr, Cr, seed=0)
    feats = [feat0.to(dev).permute(0, 2, 3, 4, 1).contiguous() for _ in range(nbuf)]
    denss = [dens0.to(dev).contiguous() for _ in range(nbuf)]
    _, extr, _ = syn.orbit_cameras(V_OUT, 1.5, 10.0)
    K = syn.intrinsics(256) / 2.0
    cam = torch.cat([extr[:, :3, :3].reshape(V_OUT, 9), extr[:, :3, 3], K[0, 0].expand(V_OUT, 1), K[1, 1].expand(V_OUT, 1),
                     K[0, 2].expand(V_OUT, 1), K[1, 2].expand(V_OUT, 1)], dim=1).repeat(B, 1).contiguous().to(dev)
    v2v = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(V_OUT).contiguous()
    of = torch.empty(V, 128, 128, Cr, device=dev)
    oo = torch.empty(V, 128, 128, device=dev)
    h = 0.5 * (Dr - 1) / Dr
    it[0] = 0

    def ren():
        k = it[0] % nbuf
        it[0] += 1
        _lib.check(lib.forge_render_fwd(_lib.ptr(feats[k]), _lib.ptr(denss[k]), _lib.ptr(cam), _lib.ptr(v2v), _lib.ptr(of), _lib.ptr(oo), None,
                                        V, B, Cr, Dr, Dr, Dr, 128, 128, 64, 0.5, 2.0, h, h, h, st), "render")
    ms = time_kernel(ren, iters=max(8, 4 * nbuf), warm=max(2, nbuf))
    byts = vol_bytes + V * 17 * 128 * 128 * 4
    taps = V * 128 * 128 * 64 * 17 * 8
    out["render_fwd_kernel"] = {"bound": "hbm", "ms": ms, "bytes": byts, "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS, "working_set_mb": nbuf * vol_bytes / 2 ** 20,
                                "gather_Gtaps_per_s": taps / ms / 1e6, "views_per_s_kernel_only": V / ms * 1e3,
                                "traffic": pmc_traffic("render_fwd_kernel")}
    del feats, denss
    # dense stage: the fp32-MFMA implicit-GEMM conv at the three ConvGRU shapes (D^3 grid, 3x3x3 taps)
    from forge_amd import convops as co
    M, Cc = B * D ** 3, 128
    x = torch.randn(M, Cc, device=dev)
    hbuf = torch.randn(M, Cc, device=dev)
    zbuf = torch.rand(M, Cc, device=dev)
    o1, o2 = torch.empty(M, Cc, device=dev), torch.empty(M, Cc, device=dev)
    grid, ig = (B, D, D, D), (D, D, D)
    for name, Cout, C2, epi in (("convgru_gates N=256 K=6912", 256, Cc, co.EPI_GRU_GATES), ("convgru_state N=128 K=6912", 128, Cc, co.EPI_GRU_OUT),
                                ("fusion_conv N=128 K=3456", 128, 0, co.EPI_AFFINE_ACT)):
        wp = torch.randn(27, Cout, Cc + C2, device=dev) * 0.01
        bias = torch.zeros(Cout, device=dev)
        sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        ms = time_kernel(lambda: co.conv_igemm(x, Cc, Cc, hbuf if C2 else None, C2, C2, wp, bias, sc, sh, 0.01, None, hbuf, zbuf, o1,
                                               o2 if epi == co.EPI_GRU_GATES else None, grid, ig, Cout, Cc if epi == co.EPI_GRU_GATES else Cout,
                                               co.TAPS_3x3x3, epilogue=epi), iters=10, warm=2)
        flops = 2.0 * M * Cout * 27 * (Cc + C2)
        out["conv_igemm " + name] = {"bound": "mfma", "ms": ms, "flops": flops, "achieved": flops / ms / 1e9,
                                                  "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / FP32_MFMA_PEAK_TF,
                                                  "used_by": "the direct form of the same convolution (`convops.winograd(False)`, odd grids, operands "
                                                          "beyond the buffer range); "
                                                             "inference, refinement and training run the Winograd launches below"}
    # the same kernel as the fusion's inference path launches it: 16 Winograd point GEMMs per launch, 3 depth taps, K = 3 Cin
    R = B * D * (D // 2) * (D // 2)
    V1, V2 = torch.randn(16, R, Cc, device=dev), torch.randn(16, R, Cc, device=dev)
    Mm = torch.empty(16, R, 2 * Cc, device=dev)
    for name, Cout, C2 in (("convgru_gates N=256 K=768", 256, Cc), ("convgru_state N=128 K=768", 128, Cc), ("fusion_conv N=128 K=384", 128, 0)):
        U = torch.randn(16, 3, Cout, Cc + C2, device=dev) * 0.01
        mm = Mm.view(-1)[:16 * R * Cout].view(16, R, Cout)
        ms = time_kernel(lambda: co.wino_gemm(V1, Cc, V2 if C2 else None, C2, U, mm, B, D, D // 2, D // 2, Cout), iters=10, warm=2)
        flops = 2.0 * 16 * R * Cout * 3 * (Cc + C2)
        out["wino_gemm " + name] = {"bound": "mfma", "ms": ms, "flops": flops, "achieved": flops / ms / 1e9, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                    "frac": flops / ms / 1e9 / FP32_MFMA_PEAK_TF, "direct_equivalent_tflops": 2.25 * flops / ms / 1e9,
                                    "kernel": "conv_igemm_kernel (16 batched 3-tap problems)"}
    return out
