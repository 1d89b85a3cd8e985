#!/usr/bin/env python
"""bench.py - rendered views/sec of the FORGE reconstruction hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path a1..a7 over one batch of B synthetic scenes
per GPU: 5 input views 256^2 -> ResNet lift -> 32^3x128 feature volumes -> HIP pose warp -> ConvGRU
fusion -> heads -> 64^3 (16+1)-channel volume -> HIP ray-march of 5 views x 128^2 rays x 64 samples ->
conv_rgb -> 5 RGB 256^2 views + masks, through forge_amd.model.FORGE.forward (GT poses, eval-mode BN,
fp32). Inputs are resident in HBM before the timed region. Weak scaling: every rank processes its own
B scenes, no data-path collective (scenes are independent); rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline        the dominant kernel of the step (conv_igemm_kernel, fp32 MFMA): FLOPs its launches EXECUTE / their HIP-event time
                  vs the 157.3 TF pipe, plus the whole step against its own executed-FLOP time floor (floor_ms, step_over_floor)
  extra_configs   (N = 1) the other BASELINE configurations, bounded: configs[2] (8 scenes), the 128^3-voxel grid, FORGE_poseEstimator3D
                  inference, one GT-pose training step, one pose-refinement iteration - each with ms_per_step, views_per_s and its
                  executed-FLOP floor
  strong_scaling  8 scenes in total split over the N ranks (the default line is weak scaling)
  kernels         per hand-written HIP kernel: algorithmic bytes / avg launch duration vs HBM peak
  stages_ms       HIP-event split of one eager step (includes host launch gaps); stages_ms_replay: each stage captured into its own hipGraph and replayed
  single_stream   the same step replayed on ONE stream, back to back (step latency; `value` keeps --pipeline-depth steps in flight)
  cpu_baseline    the CPU oracle (reference semantics, torch-CPU) timed on this box's host cores on a
                  bounded sample of the same workload (N=1, rank 0 only)
  ranks_ok        ranks that completed the timed region (a failing rank reports its error instead of hanging the others)

`--train`: the data-parallel TRAINING step instead (BASELINE configs[3]: FORGE_poseEstimator3D under SyncBatchNorm + DDP, forward +
backward + clip + Adam, RCCL gradient all-reduce) - an extra mode with its own metric string, never the driver's line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the HSA runtime starts (first torch.cuda call): this host driver supports dmabuf IPC only - a rank launched by somebody else's
# torch.distributed.run (not through rank_env below) must see it too, or RCCL's cross-process buffer exchange fails
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from forge_amd import _lib, dist as fdist, synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32 dense peak
T_IN, V_OUT = 5, 5
# algorithmic work per scene
GF_ENCODER = 64.3 * T_IN
GF_FUSE = 927.7
GF_HEADS = 45.3
GF_CONVRGB = 0.80 * V_OUT


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
        rec.setdefault("conv_igemm_kernel<%s>" % co.TILE_NAMES[co.wino_gemm_tile(R, Cout, C1 + C2)], []).append((e0, e1, 2.0 * 16 * R * Cout * U.shape[1] * (C1 + C2), (16 * R, Cout, U.shape[1], C1 + C2), 2.25))
        return out

    def input_timed(x, C, ld, n, D, H, W, **kw):
        e0, e1 = ev()
        e0.record()
        out = o_i(x, C, ld, n, D, H, W, **kw)
        e1.record()
        rec.setdefault("wino_input_kernel", []).append((e0, e1, 4.0 * n * D * H * W * C * (kw.get("nsum", 1) + 4)))   # reads the rows (of nsum views) once, writes 16 points x R = 4x
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
        rec.setdefault("wino_output_kernel", []).append((e0, e1, 4.0 * rows * (4 * Cout + side)))       # reads 16 points x R x Cout = 4x, then the tail's operands
        return r
    co.wino_gemm, co.wino_input, co.wino_output = gemm_timed, input_timed, output_timed
    undo.append(lambda: (setattr(co, "wino_gemm", o_g), setattr(co, "wino_input", o_i), setattr(co, "wino_output", o_o)))
    return rec, undo


def pmc_traffic(prefix):
    """HBM bytes per launch of the kernel whose summary key contains `prefix`, from the committed rocprofv3 PMC passes (profiles/*pmc_summary.json: separate --pmc
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


KLOOP_CEILING_TF = {"64x64": 130.0, "64x128": 137.0, "128x128": 141.0}     # LDS -> MFMA loop alone (no global -> LDS staging), direct gates launch K = 6912:
                                                                            # debug builds of tools/debug/gemm_ceiling.py, profiles/TUNING_LOG.md "K-loop ceiling"


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


def time_kernel(fn, iters=20, warm=3):
    """Average duration (ms) of one launch of `fn`, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


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
                                                  "used_by": "the direct form of the same convolution (`convops.winograd(False)`, odd grids, operands beyond the buffer range); "
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


def physical_cores():
    """Physical cores this process may run on (unique (socket, core) pairs of /proc/cpuinfo, capped by the affinity mask)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:
        allowed = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        n = len(pairs) or allowed
    except Exception:
        n = allowed
    return max(1, min(n, allowed)), allowed


def cpu_quota_cores():
    """CPU-time budget of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable. The GPU boxes of
    this pool list 256 hardware threads but run the job under `cpu.max = 1600000 100000` = 16 cores: more runnable threads than that are
    throttled, which is what made round 3's 8 x 16-thread leg take 8x longer per forward than one process (tools/cpu_quota_probe.py)."""
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        if a != "max":
            return float(a) / float(b)
    except Exception:
        pass
    try:
        q, p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / float(p_)
    except Exception:
        pass
    return None


def _spin(seconds, q):
    t0, n, x = time.perf_counter(), 0, 1
    while time.perf_counter() - t0 < seconds:
        for _ in range(20000):
            x = (x * 1103515245 + 12345) & 0x7fffffff
        n += 20000
    q.put(n)


def effective_parallelism(ks, seconds=0.5):
    """Aggregate rate of k single-thread spin loops relative to one: what the scheduler really grants this container (a plateau = the quota)."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    out, base = {}, None
    for k in ks:
        q = ctx.Queue()
        ps = [ctx.Process(target=_spin, args=(seconds, q)) for _ in range(k)]
        t0 = time.perf_counter()
        for p_ in ps:
            p_.start()
        tot = sum(q.get() for _ in ps)
        for p_ in ps:
            p_.join()
        rate = tot / (time.perf_counter() - t0)
        base = base or rate
        out[str(k)] = round(rate / base, 2)
    return out


def _cpu_forward_fn(seed, threads):
    """(run, ref-holder) of one oracle hot-path forward of ONE seeded scene on `threads` torch-CPU threads."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import forge_oracle as fo
    from forge_amd.model import FORGE
    cfg = syn.kubric_config()
    weights = syn.seeded_state_dict(FORGE(cfg).state_dict(), 0)
    one = syn.make_sample(1, T_IN, 256, 1.5, seed=seed)
    torch.set_num_threads(threads)

    def run():
        with torch.no_grad():
            return fo.forward_hot_path(one["images"], one["cam_poses_cv2_canonicalized"], one["cam_extrinsics_cv2_canonicalized"],
                                       one["K_cv2"], weights, cfg, order_by_distance=True)
    return run


def core_sets(nsets, per_set):
    """nsets disjoint sets of per_set logical CPUs, one hardware thread per physical core, consecutive cores of one socket together."""
    cores, phys, core, proc = {}, None, None, None
    try:
        allowed = os.sched_getaffinity(0)
        for line in list(open("/proc/cpuinfo")) + [""]:
            if line.startswith("processor"):
                proc = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if proc is not None and proc in allowed and phys is not None:
                    cores.setdefault((phys, core), proc)
                phys = core = proc = None
    except Exception:
        return None
    order = [cores[k] for k in sorted(cores)]
    if len(order) < nsets * per_set:
        return None
    return [order[i * per_set:(i + 1) * per_set] for i in range(nsets)]


def cpu_worker(threads, n_forward, seed):
    """`bench.py --cpu-worker THREADS N SEED`: one process of the scene-parallel CPU baseline. Its CPU set was applied by the parent
    BEFORE exec (preexec_fn -> sched_setaffinity), so the OpenMP runtime sizes and places its threads inside that set; no OMP_PROC_BIND
    (round 2 set OMP_PROC_BIND=close with the affinity applied after `import torch`: the OpenMP places had already been computed from the
    full mask, every process bound its 16 threads to the SAME first cores - 35 s per 1.1 s forward). Prints 'CPUWORKER t0 t1 n'."""
    run = _cpu_forward_fn(seed, threads)
    run()                                            # warm-up (allocator, oneDNN primitive caches)
    print("CPUWORKER_READY", flush=True)
    sys.stdin.readline()                             # start line from the parent: all workers begin their timed forwards together
    t0 = time.time()
    for _ in range(n_forward):
        run()
    print("CPUWORKER %.6f %.6f %d" % (t0, time.time(), n_forward), flush=True)


def cpu_baseline(sample, weights, cfg):
    """The oracle (reference semantics, torch-CPU fp32: the port of the reference's CPU path) on this box's host cores, on a BOUNDED
    sample (one scene per forward; ~20-40 s of CPU work in total).
      1. single process: every candidate thread count gets 1 warm-up + 1 timed forward (a count whose warm-up exceeds 3 s is
         recorded as such and not timed again), then the fastest count gets 5 timed forwards;
      2. scene-parallel: P processes x T threads = all physical cores, each process running its own scene (how a CPU deployment would
         fill the box; torch-CPU convolutions do not scale past ~16-32 threads), 1 warm-up + 1 timed forward each, started together.
    `value` is the better of the two aggregates; both are reported."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import forge_oracle as fo
    phys_listed, hw = physical_cores()
    quota = cpu_quota_cores()
    # the cores this job can actually USE: the cgroup CPU-time quota when there is one (threads beyond it are throttled, not run)
    phys = max(1, min(phys_listed, int(quota))) if quota else phys_listed
    one = {k: v[:1].cpu() for k, v in sample.items()}

    def run():
        with torch.no_grad():
            return fo.forward_hot_path(one["images"][:, :T_IN], one["cam_poses_cv2_canonicalized"][:, :T_IN],
                                       one["cam_extrinsics_cv2_canonicalized"][:, :T_IN], one["K_cv2"][:, :T_IN],
                                       weights, cfg, order_by_distance=True)
    cands = sorted({c for c in (4, 8, 16, 32, 64, phys) if 1 <= c <= phys})
    sweep, ref = {}, None
    for nt in cands:
        torch.set_num_threads(nt)
        t0 = time.time()
        r = run()
        warm = time.time() - t0
        ref = r if ref is None else ref
        if warm > 3.0 and sweep:                     # hopeless thread count (3-4x slower than the best so far): its warm-up is its record
            sweep[nt] = {"warmup_s": round(warm, 2), "timed_s": None}
            continue
        t1 = time.time()
        run()
        sweep[nt] = {"warmup_s": round(warm, 2), "timed_s": round(time.time() - t1, 3)}
    best_nt = min((v["timed_s"] if v["timed_s"] is not None else v["warmup_s"], k) for k, v in sweep.items())[1]
    torch.set_num_threads(best_nt)
    run()
    times = []
    for _ in range(5):
        t0 = time.time()
        run()
        times.append(time.time() - t0)
    single = {"threads": best_nt, "timed_forwards": 5, "s_per_forward": sum(times) / 5, "views_per_s": V_OUT * 5 / sum(times)}
    # scene-parallel over all USABLE cores: processes x threads = the budget (8 threads per process: the oracle's convolutions scale to ~8)
    tpp = min(8, phys)
    nproc = max(1, phys // tpp)
    nfw = 2
    par = None
    try:
        sets = core_sets(nproc, tpp)                 # each process pinned to its own 16 physical cores (one socket, no SMT siblings)
        env = {k: v for k, v in os.environ.items() if k not in ("OMP_PROC_BIND", "OMP_PLACES", "GOMP_CPU_AFFINITY", "KMP_AFFINITY")}
        env.update(OMP_NUM_THREADS=str(tpp), MKL_NUM_THREADS=str(tpp))

        def pin(cpus):                               # runs in the child between fork and exec: the interpreter starts inside its CPU set
            return (lambda: os.sched_setaffinity(0, set(cpus))) if cpus else None
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(tpp), str(nfw), str(2000 + i)],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env,
                                  preexec_fn=pin(sets[i] if sets else None)) for i in range(nproc)]
        for p in procs:
            while True:
                line = p.stdout.readline()
                if not line or line.startswith("CPUWORKER_READY"):
                    break
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        spans = []
        for p in procs:
            out, _ = p.communicate(timeout=300)
            for line in out.splitlines():
                if line.startswith("CPUWORKER "):
                    a, b, n = line.split()[1:]
                    spans.append((float(a), float(b), int(n)))
        if len(spans) == nproc:
            wall = max(b for _, b, _ in spans) - min(a for a, _, _ in spans)
            par = {"processes": nproc, "threads_per_process": tpp, "pinned": bool(sets), "timed_forwards": nproc * nfw, "wall_s": wall,
                   "views_per_s": V_OUT * sum(n for _, _, n in spans) / wall}
    except Exception as e:                                              # the single-process number stands
        par = {"error": repr(e)}
    use_par = bool(par) and par.get("views_per_s", 0.0) > single["views_per_s"]
    value = par["views_per_s"] if use_par else single["views_per_s"]
    cores = par["processes"] * par["threads_per_process"] if use_par else best_nt
    lscpu = ""
    try:
        lscpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    try:
        eff = effective_parallelism([1, 8, 16, 32] if hw >= 32 else [1, max(1, hw // 2), hw])
    except Exception as e:
        eff = {"error": repr(e)}
    return {"value": value, "unit": "views/s", "cores": cores, "physical_cores": phys_listed, "usable_cores": phys, "cgroup_cpu_quota_cores": quota,
            "host_hw_threads": hw, "effective_parallelism": eff, "cpu_model": lscpu, "kind": "port",
            "note": "cores = the threads that produced `value`. The box lists %d physical cores / %d hardware threads, but the job runs under a cgroup "
                    "CPU-time quota of %s cores (effective_parallelism: aggregate rate of k spin loops / one - it plateaus at the quota), so the "
                    "baseline is sized to the quota; a leg with more runnable threads than that is throttled, not faster" % (phys_listed, hw, quota),
            "sample": "oracle hot path, 1 scene per forward (5x256^2 in, 32^3/64^3 grids, 5x128^2x64 rays out), torch-CPU fp32; "
                      "thread sweep %s; single process: 1 warm-up + 5 timed forwards at %d threads; scene-parallel: %s"
                      % (sorted(sweep), best_nt, ("%d processes x %d threads, 1 warm-up + %d timed forwards each" % (nproc, tpp, nfw))),
            "thread_sweep": sweep, "single_process": single, "scene_parallel": par}, ref


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: re-exec under torch.distributed.run with N ranks on this node
    (one process per GPU; rendezvous on 127.0.0.1). Returns only in the children / for N = 1."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=rank_env(os.environ)))


def rank_env(base):
    """Environment of the ranks: dmabuf IPC for RCCL across processes on this driver, a bounded OpenMP pool per rank, RCCL warnings on."""
    env = dict(base)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("NCCL_DEBUG", "WARN")
    return env


def pin_rank_to_gpu_numa(dev):
    """Bind this rank's host threads to the CPUs of its GPU's NUMA node (PCI bus id -> /sys/bus/pci/devices/<bdf>/numa_node ->
    /sys/devices/system/node/nodeN/cpulist): launch latency and pinned-memory copies stay on the GPU's socket. Best effort: returns a
    description for the JSON line, never raises."""
    try:
        prop = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), prop.pci_bus_id, prop.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "pinned": False, "reason": "no NUMA information"}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pci": bdf, "numa_node": node, "pinned": False, "reason": "node CPUs outside the allowed set"}
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception as e:                                        # containers without sysfs, exotic topologies: run unpinned
        return {"pinned": False, "reason": repr(e)[:120]}


def dry_run(args, rank, world):
    """`--dry-run`: the launch / rendezvous / timing-reduction skeleton of this entry point on CPU over gloo, with a token CPU workload
    instead of the HIP step (tests/test_dist_cpu.py runs `python bench.py --gpus 8 --dry-run` here, where there is no GPU). With --train the
    token workload is a DistributedDataParallel step (bucketed gradient all-reduce over gloo), as the real --train mode wraps the model."""
    fdist.init(backend="gloo")
    fdist.barrier()
    ddp = opt = None
    if args.train:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
        ddp = torch.nn.parallel.DistributedDataParallel(net) if world > 1 else net
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    ok, err = 1.0, None
    t0 = time.perf_counter()
    acc = 0.0
    try:
        for _ in range(args.steps):
            if ddp is not None:
                opt.zero_grad()
                loss = ddp(torch.full((4, 64), 1.0 + rank)).square().mean()
                loss.backward()
                opt.step()
                acc += float(loss)
            else:
                acc += float(torch.ones(64, 64).sum())
    except Exception as e:                                        # a failing rank still joins the reductions below
        ok, err = 0.0, repr(e)
    fdist.barrier()
    dt = fdist.all_reduce_scalars([time.perf_counter() - t0], "cpu", "max")[0]
    units, ranks_ok = fdist.all_reduce_scalars([float(args.scenes * (10 if args.train else V_OUT) * args.steps), ok], "cpu", "sum")
    same = None
    if ddp is not None and world > 1:                             # DDP keeps the replicas identical: the parameter checksum agrees on all ranks
        chk = float(sum(p.detach().double().sum() for p in ddp.parameters()))
        lo, hi = fdist.all_reduce_scalars([chk], "cpu", "min")[0], fdist.all_reduce_scalars([chk], "cpu", "max")[0]
        same = abs(hi - lo) < 1e-9 * max(1.0, abs(hi))
    multi = None
    if world > 1 and not args.train:
        # the sub-record skeleton of the real multi-rank line (multi_rank_records: watchdog, per-record try block, error gathering) with token
        # workloads: a DDP step timed with and without no_sync(), a record that FAILS on the last rank (reported, the others carry on), and the
        # differentiable ray-sharded render (all_gather forward, all-reduce backward) on a toy render function
        def token_ddp():
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
            dd = torch.nn.parallel.DistributedDataParallel(net)
            o = torch.optim.Adam(net.parameters(), lr=1e-3)

            def st():
                o.zero_grad()
                dd(torch.full((4, 64), 1.0 + rank)).square().mean().backward()
                o.step()

            def st_ns():
                with dd.no_sync():
                    st()
            a, b_ = _bracketed(st, 2, 1, "cpu"), _bracketed(st_ns, 2, 1, "cpu")
            return {"ms_per_step": a * 1e3, "ms_per_step_no_sync": b_ * 1e3, "gradient_bytes_all_reduced_per_step": sum(p.numel() for p in net.parameters()) * 4}

        def token_fail():
            if rank == world - 1:
                raise RuntimeError("rehearsed failure on rank %d" % rank)
            return {"ms_per_step": 0.0}

        def token_rays():
            Hr = 2 * world
            feat = torch.ones(1, 2, 2, 2, 2, requires_grad=True)
            dens = torch.ones(1, 1, 2, 2, 2, requires_grad=True)
            cam = torch.zeros(3, 16)
            toy = lambda f, d, c, v2v, hr, wr, *a: (f.sum() * torch.ones(3, 2, hr, wr) + c[:, 15].reshape(3, 1, 1, 1), d.sum() * torch.ones(3, 1, hr, wr))   # noqa: E731
            o = fdist.render_rays_sharded(feat, dens, cam, None, Hr, 4, 8, 0.5, 2.0, (1.0, 1.0, 1.0), render_fn=toy)
            (o[0].sum() + o[1].sum()).backward()
            return {"rows": int(o[0].shape[2]), "d_feat": float(feat.grad.sum()), "expected_d_feat": float(3 * 2 * Hr * 4 * 16)}
        def token_hang():
            if rank == world - 1:
                time.sleep(3600)                                         # a rank stuck for ever; the others block in the next collective
            return {}
        recs = (("ddp_train", token_ddp), ("failing_record", token_fail), ("ray_sharded_joint", token_rays))
        multi = multi_rank_records(args, rank, world, "cpu", {"metric": "dry run", "value": None, "n_gpus": world, "dry_run": True},
                                   records=(("hang", token_hang),) if args.rehearse_hang else recs)
    if rank == 0:
        print(json.dumps({"metric": "rendered views/sec (5 views, 128^2 px, 64^3 voxel)", "value": None, "unit": "views/s", "n_gpus": world, "multi_rank": multi,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "views_counted": units, "ms_per_step": dt / args.steps * 1e3,
                          "scaling": "weak", "ranks_ok": int(ranks_ok), "process_group": fdist.group_info(), "train": bool(args.train), "replicas_identical": same, "error": err,
                          "config": {"workload": "dry run: no HIP work, launch + rendezvous + reductions only"}}), flush=True)
    fdist.barrier()
    fdist.shutdown()


def floor_of(gflop, ms):
    """A step against its own executed-FLOP time floor on the fp32 MFMA pipe."""
    floor_ms = gflop / FP32_MFMA_PEAK_TF
    return {"executed_gflop": gflop, "floor_ms": floor_ms, "executed_frac": floor_ms / ms if ms > 0 else None, "step_over_floor": ms / floor_ms if floor_ms > 0 else None}


def _timed(fn, steps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def timed_region(fn, steps, warmup, repeats=1):
    """W untimed steps, then `repeats` regions of EXACTLY K timed steps, each between (barrier, synchronize) pairs; a rank that fails keeps
    the barrier count. Returns (ok, error, last output, [seconds per region])."""
    good, msg, out, dts = 1.0, None, None, []
    try:
        for _ in range(warmup):
            out = fn()
        torch.cuda.synchronize()
    except Exception as e:
        good, msg = 0.0, repr(e)[:400]
    for _ in range(max(1, repeats)):
        fdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            if good:
                for _ in range(steps):
                    out = fn()
                torch.cuda.synchronize()
        except Exception as e:
            good, msg = 0.0, repr(e)[:400]
        fdist.barrier()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    return good, msg, out, dts


def region_stats(dts_max, steps, units_per_step):
    """dts_max = every region's duration (max over ranks): `value` = units / MEDIAN region; the spread of the same run beside it."""
    import statistics
    med = statistics.median(dts_max)
    per = lambda d: units_per_step * steps / d if d > 0 else None
    return med, {"regions": len(dts_max), "steps_per_region": steps, "value_median": per(med), "value_min": per(max(dts_max)), "value_max": per(min(dts_max)),
                 "ms_per_step_median": med / steps * 1e3, "ms_per_step_min": min(dts_max) / steps * 1e3, "ms_per_step_max": max(dts_max) / steps * 1e3}


def extra_configs(dev, steps=5):
    """The other BASELINE configurations on this GPU, bounded (<= `steps` timed steps each), AFTER the headline timed region (N = 1):
    configs[2] (8 scenes), the 128^3-voxel grid (n1 / configs[3]-[4] grid), FORGE_poseEstimator3D inference, one GT-pose training step
    (configs[3] per-GPU step at the reference-native grid) and one pose-refinement iteration (row f2). Each entry: workload, ms_per_step,
    views_per_s and `roofline` = the FLOPs the step's matrix-core launches execute (FlopMeter around one eager pass) as a time floor on the
    157.3 TF fp32 MFMA pipe. A configuration that fails reports its error and the others still run."""
    from forge_amd import geo_utils, refine
    from forge_amd.flopmeter import FlopMeter
    from forge_amd.graph import GraphedCall, GraphedForward, PipelinedForward
    from forge_amd.model import FORGE
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
    from forge_amd import train
    from forge_amd.train import grouped_mse
    out = []
    holder = {}
    ds = syn.SyntheticDataset(1.5)
    cfg = syn.kubric_config()

    def build(cls, train=False):
        m = cls(cfg)
        m.load_state_dict(syn.seeded_state_dict(m.state_dict(), 0))
        m = m.to(dev)
        return m.train() if train else m.eval()

    def entry(name, workload, views, fn_eager, fn_timed, n=steps, make_pipe=None):
        try:
            with FlopMeter() as fm:
                fn_eager()
            torch.cuda.synchronize()
            ms = _timed(fn_timed, n)
            e = {"name": name, "workload": workload, "steps": n, "ms_per_step": ms, "views_per_s": views / ms * 1e3,
                 "roofline": dict(floor_of(fm.gflop, ms), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm.gflop / ms, launches=fm.launches)}
            if make_pipe is not None:                          # the same step with several replays in flight (PipelinedForward), as the headline runs it
                holder.clear()
                torch.cuda.empty_cache()
                pipe, depth = make_pipe()
                msp = _timed(pipe, 2 * n, warm=depth)
                e["pipelined"] = dict(floor_of(fm.gflop, msp), depth=depth, ms_per_step=msp, views_per_s=views / msp * 1e3)
                del pipe
            out.append(e)
        except Exception as e:
            out.append({"name": name, "workload": workload, "error": repr(e)[:300]})
        torch.cuda.empty_cache()

    model = build(FORGE)
    # --- configs[2]: 8 scenes per GPU
    s8 = {k: v.to(dev) for k, v in syn.make_sample(8, T_IN, 256, 1.5, seed=1000).items()}

    def eager8():
        with torch.no_grad():
            model(s8, ds, dev)

    def timed8():
        if "g" not in holder:
            holder["g"] = GraphedForward(model, s8, ds, dev)
        holder["g"](s8)
    def pipe8():
        p = PipelinedForward(model, s8, ds, dev, depth=2, warmup=1)
        return (lambda: p(s8)), 2
    entry("configs[2]", "BASELINE configs[2]: FORGE hot path, 8 scenes/GPU x 5 views -> 40 rendered views per step (hipGraph replay)", 40, eager8, timed8,
          make_pipe=pipe8)
    holder.clear()
    del s8
    # --- 128^3-voxel scenes (synthetic 64^3 feature volumes through reconstruct)
    s1 = {k: v.to(dev) for k, v in syn.make_sample(1, T_IN, 256, 1.5, seed=1000).items()}
    gen = torch.Generator(device=dev).manual_seed(77)
    f64 = torch.randn(1, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    p64 = s1["cam_poses_cv2_canonicalized"][:, :T_IN].contiguous()
    c64 = geo_utils.camera_dict(s1["cam_extrinsics_cv2_canonicalized"][:, :V_OUT], s1["K_cv2"][:, :V_OUT])

    def eager64():
        with torch.no_grad():
            return model.reconstruct(f64, p64, c64)[:2]

    def timed64():
        if "g" not in holder:
            holder["g"] = GraphedCall(eager64, dev)
        holder["g"]()
    entry("grid64", "128^3-voxel scenes (configs[3]/[4] grid): 1 scene x 5 synthetic [128,64^3] feature volumes -> rotate(D=64) -> fusion at "
          "M=262144 -> heads -> 128^3 x 17 volume -> 5 views (hipGraph replay)", 5, eager64, timed64)
    holder.clear()
    del f64
    # --- pose refinement iteration (row f2): t = 5 views, 4 free poses, hipGraph replay inside refine_poses
    try:
        with torch.no_grad():
            feats = model.encoder_3d.get_feat3D(s1["images"][0, :T_IN]).reshape(1, T_IN, 128, 32, 32, 32)
            gt7 = geo_utils.mat2quat(s1["cam_poses_rel_cv2"][0, 1:T_IN])
            tgt_i, tgt_m, _, _, _ = refine._render_views(model, cfg, ds, feats, gt7, s1["K_cv2"][:, :T_IN], dev)
        init = gt7.clone()
        init[:, 4:] += 0.02
        with FlopMeter() as fm:                                    # eager iterations only (one here): forward + data-gradient backward
            refine.refine_poses(model, cfg, ds, feats, init, tgt_i, tgt_m, s1["K_cv2"][:, :T_IN], dev, iter_num=0, use_graph=False)
        _, _, dt = refine.refine_poses(model, cfg, ds, feats, init, tgt_i, tgt_m, s1["K_cv2"][:, :T_IN], dev, iter_num=2 * steps + 3, use_graph=True)
        ms = dt * 1e3
        out.append({"name": "refinement", "workload": "pose-refinement iteration (kubric_eval.py:412-530): 1 scene, 5 views, 4 free 7-D poses; rotate -> fuse "
                    "-> heads -> ray-march -> conv_rgb forward + data-gradient backward + Adam, hipGraph replay", "steps": 2 * steps,
                    "ms_per_step": ms, "views_per_s": T_IN / ms * 1e3,
                    "roofline": dict(floor_of(fm.gflop, ms), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm.gflop / ms, launches=fm.launches)})
        try:                                                       # two refinement problems in flight (refine_poses_many): per iteration AND instance
            probs = [(feats, init, tgt_i, tgt_m, s1["K_cv2"][:, :T_IN]), (feats, init.clone(), tgt_i, tgt_m, s1["K_cv2"][:, :T_IN])]
            _, dt2 = refine.refine_poses_many(model, cfg, ds, probs, dev, iter_num=2 * steps, depth=2)
            out[-1]["pipelined"] = dict(floor_of(fm.gflop, dt2 * 1e3), depth=2, ms_per_step=dt2 * 1e3, views_per_s=T_IN / dt2)
        except Exception as e:
            out[-1]["pipelined"] = {"error": repr(e)[:200]}
    except Exception as e:
        out.append({"name": "refinement", "error": repr(e)[:300]})
    del model
    torch.cuda.empty_cache()
    # --- FORGE_poseEstimator3D inference: three fusions, 10 rendered views per scene
    m3 = build(FORGE_poseEstimator3D)

    def eager3():
        with torch.no_grad():
            m3(s1, ds, dev)

    def timed3():
        if "g" not in holder:
            holder["g"] = GraphedForward(m3, s1, ds, dev)
        holder["g"](s1)
    def pipe3():
        p = PipelinedForward(m3, s1, ds, dev, depth=4, warmup=1)
        return (lambda: p(s1)), 4
    entry("pose3d_inference", "FORGE_poseEstimator3D inference (GT poses): 1 scene x 5 views -> 3 fusions (shared input halves) -> 10 rendered views "
          "(hipGraph replay)", 10, eager3, timed3, make_pipe=pipe3)
    holder.clear()
    # --- FORGE with PREDICTED poses in inference (kubric_eval.py:371-410 predict_initial / demo.py): both pose estimators + pose head -> cameras -> reconstruction -> 10 views
    try:
        mj = FORGE(syn.kubric_config(use_gt_pose=False, parameter="joint"))
        mj.load_state_dict(syn.seeded_state_dict(mj.state_dict(), 0))
        mj = mj.to(dev).eval()
        s10 = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}

        def eagerj():
            with torch.no_grad():
                mj(s10, ds, dev)

        def timedj():
            if "g" not in holder:
                holder["g"] = GraphedForward(mj, s10, ds, dev)
            holder["g"](s10)
        entry("joint_inference", "FORGE inference with PREDICTED poses (2-D + 3-D pose estimators + pose head -> cameras): 1 scene x 5 input views -> 10 rendered views "
              "(5 predicted + 5 given novel cameras); the 2-D estimator on a side HIP stream beside the encoder (hipGraph replay)", 10, eagerj, timedj)
        holder.clear()
        del mj, s10
    except Exception as e:
        out.append({"name": "joint_inference", "error": repr(e)[:300]})
    torch.cuda.empty_cache()
    # --- one GT-pose training step (configs[3] per-GPU step at the reference-native 32^3 / 64^3 grids): forward + backward + clip + Adam, eager
    m3 = m3.train()
    opt = torch.optim.Adam([p for p in m3.parameters() if p.requires_grad], lr=1e-4, fused=True)     # torch's multi-tensor Adam: same update, one launch chain

    def train_step():
        imgs, masks = m3(s1, ds, dev)
        mi = grouped_mse(imgs.reshape(1, 10, 3, 256, 256), s1["images"][:, :T_IN], T_IN)
        mm = grouped_mse(masks.reshape(1, 10, 1, 256, 256), s1["fg_probabilities"][:, :T_IN], T_IN)
        loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        train.clip_grad_norm_(m3.parameters(), 10.0)
        opt.step()
    entry("train_step", "GT-pose training step (kubric_train_pose_3D.py; scripts/kubric_trainer.py:47-59): FORGE_poseEstimator3D, 1 scene x 5 views, "
          "3 fusions, 10 rendered views, fused MSE, backward, clip 10, Adam; train-mode BatchNorm on the HIP kernels; eager launch", 10, train_step, train_step)
    # the per-GPU shape of BASELINE configs[3]: 4 scenes per GPU (bounded: 3 timed steps of ~175 ms)
    try:
        s4 = {k: v.to(dev) for k, v in syn.make_sample(4, T_IN, 256, 1.5, seed=1001).items()}

        def train_step4():
            imgs, masks = m3(s4, ds, dev)
            mi = grouped_mse(imgs.reshape(4, 10, 3, 256, 256), s4["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(4, 10, 1, 256, 256), s4["fg_probabilities"][:, :T_IN], T_IN)
            loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            train.clip_grad_norm_(m3.parameters(), 10.0)
            opt.step()
        entry("train_step_4_scenes", "the same training step at configs[3]'s per-GPU batch: 4 scenes x 5 views -> 40 rendered views per step; eager launch",
              40, train_step4, train_step4, n=steps)
        del s4
    except Exception as e:
        out.append({"name": "train_step_4_scenes", "error": repr(e)[:300]})
    # BASELINE configs[3] at its REAL per-GPU shape: 4 scenes x 128^3-voxel render grid = 64^3 feature grid (models/rotate.py:115-117; the encoder cannot
    # produce 64^3 features from 256^2 images, models/encoder.py:49, so synthetic [4,5,128,64^3] feature volumes enter at rotate): rotate(D=64), three
    # fusions at M = 4 x 262144, heads to 128^3, 40 ray-marched views, loss, backward (data + weight gradients), clip, Adam
    try:
        s4 = {k: v.to(dev) for k, v in syn.make_sample(4, T_IN, 256, 1.5, seed=1001).items()}
        gen4 = torch.Generator(device=dev).manual_seed(78)
        f4 = torch.randn(4, T_IN, 128, 64, 64, 64, device=dev, generator=gen4).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
        c4 = geo_utils.camera_dict(s4["cam_extrinsics_cv2_canonicalized"][:, :T_IN].repeat(1, 2, 1, 1), s4["K_cv2"][:, :T_IN].repeat(1, 2, 1, 1))
        p4 = s4["cam_poses_cv2_canonicalized"][:, :T_IN].contiguous()

        def train_step4g():
            imgs, masks = m3.reconstruct(f4, p4, c4)[:2]
            mi = grouped_mse(imgs.reshape(4, 10, 3, 256, 256), s4["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(4, 10, 1, 256, 256), s4["fg_probabilities"][:, :T_IN], T_IN)
            loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            train.clip_grad_norm_(m3.parameters(), 10.0)
            opt.step()
        entry("train_step_4_scenes_grid64", "BASELINE configs[3] per-GPU shape: GT-pose training step, 4 scenes x 5 synthetic [128,64^3] feature volumes "
              "(128^3-voxel render grid) -> rotate(D=64) -> 3 fusions -> heads -> 128^3 x 17 volumes -> 40 rendered views, backward, clip 10, Adam; eager launch",
              40, train_step4g, train_step4g, n=steps)
        del s4, f4
    except Exception as e:
        out.append({"name": "train_step_4_scenes_grid64", "error": repr(e)[:300]})
    torch.cuda.empty_cache()
    out.extend(joint_configs(dev, steps=max(3, steps // 2)))
    # the same step captured into ONE hipGraph (forge_amd.graph.GraphedStep: forward, loss, backward, clip, capturable Adam) - single-process
    # training is host-bound at one scene (~1000 launches per step); reported beside the eager number, which is what a DDP wrapper runs
    try:
        from forge_amd.graph import GraphedStep
        opt_g = torch.optim.Adam([p for p in m3.parameters() if p.requires_grad], lr=1e-4, capturable=True)

        def graph_fn():
            imgs, masks = m3(s1, ds, dev)
            mi = grouped_mse(imgs.reshape(1, 10, 3, 256, 256), s1["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(1, 10, 1, 256, 256), s1["fg_probabilities"][:, :T_IN], T_IN)
            loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
            loss.backward()
            train.clip_grad_norm_(m3.parameters(), 10.0)
            opt_g.step()
            return loss.detach()
        gs = GraphedStep(graph_fn, opt_g, warmup=2)
        msg = _timed(gs, steps)
        ts = [e for e in out if e.get("name") == "train_step" and "ms_per_step" in e]
        if ts:
            ts[-1]["hipgraph_replay"] = dict(floor_of(ts[-1]["roofline"]["executed_gflop"], msg), ms_per_step=msg, views_per_s=10 / msg * 1e3)
        del gs
    except Exception as e:
        ts = [x for x in out if x.get("name") == "train_step"]
        if ts:
            ts[-1]["hipgraph_replay"] = {"error": repr(e)[:200]}
    return out


def joint_stock_share():
    """Share of the joint step's kernel time spent in stock-torch (MIOpen / rocBLAS / ATen) kernels, from the committed rocprofv3 kernel trace
    of tools/joint_step_probe.py (profiles/r05_joint_*_kernel_share.json, written by tools/joint_kernel_share.py): a per-name attribution the
    process cannot make about itself. None when no profile is committed."""
    import glob
    res = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_joint_*kernel_share.json"))):
        try:
            d = json.load(open(f))
            res[d.get("workload", os.path.basename(f))] = {"stock_torch_share_of_kernel_time": d["stock_share"], "forge_share_of_kernel_time": d["forge_share"],
                                                           "kernel_ms_per_step": d.get("kernel_ms_per_step"), "source": os.path.basename(f)}
        except Exception:
            continue
    return res or None


def joint_configs(dev, steps=5):
    """BASELINE configs[4] on one GPU: the joint 2D3D fine-tune iteration of kubric_train_joint.py:111-141 - FORGE with
    PREDICTED poses (attention blocks of the 2-D / 3-D pose estimators and the pose head on stock torch kernels; encoder / rotate / fusion / heads / ray-march /
    conv_rgb and, since round 5, every convolution + BatchNorm of the two pose estimators on the HIP kernels), 5 input + 5 novel views, compute_all_loss_nvs (scripts/kubric_compute_loss.py:121-172), backward through the
    pose chain (rotate's d pose, the ray-marcher's d(R, T)), clip 10, Adam over the parameter list of kubric_train_joint.py:111-116.
      joint_step          reference-native grids (32^3 features, 64^3 render volume)
      joint_step_grid64   the configuration's 128^3-voxel scenes: synthetic [1,5,128,64^3] feature volumes enter the reconstruction
                          (FORGE.forward(features_recon=...)), the pose networks keep their native inputs
    Each entry: ms_per_step, views_per_s, `roofline` = the FLOPs libforge's matrix-core launches execute as a time floor (stock-torch FLOPs are
    counted separately by torch's FlopCounterMode and NOT part of that floor), and `stock_torch` = the pose networks' own forward + backward
    timed alone on the same inputs (live) beside the per-kernel-name share of a committed rocprofv3 trace."""
    from forge_amd import train
    from forge_amd.flopmeter import FlopMeter
    from forge_amd.model import FORGE
    out = []
    cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
    cfg.loss.regu_origin_proj = 1.0                                   # config/kubric/joint_pose_2d3d.yaml:34-38 (perceptual term: no VGG weights offline)
    ds = syn.SyntheticDataset(1.5)
    try:
        model = FORGE(cfg)
        model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
        model = model.to(dev).train()
        params = [p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head, model.render)
                  for p in m.parameters()]                            # kubric_train_joint.py:111-116
        opt = torch.optim.Adam(params, lr=1e-4, fused=True)
        sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}
        gen = torch.Generator(device=dev).manual_seed(79)
        f64 = torch.randn(1, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    except Exception as e:
        return [{"name": "joint_step", "error": repr(e)[:300]}]

    def make_step(feats, smp=None):
        call = model if feats is None else (lambda s, d, dv: model(s, d, dv, features_recon=feats))
        smp = sample if smp is None else smp

        def step():
            loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, smp, ds, call, {}, dev)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            train.clip_grad_norm_(model.parameters(), 10.0)
            opt.step()
            return loss
        return step

    def pose_nets_only():
        """both pose estimators + pose head alone: forward and backward on the step's own (detached) inputs"""
        with torch.no_grad():
            clips = sample["images"][:, :T_IN]
            feats = model.encoder_3d.get_feat3D(clips.reshape(T_IN, 3, 256, 256)).reshape(1, T_IN, 128, 32, 32, 32)
        feats = feats.detach().requires_grad_(True)

        def run():
            _, _, pose = model.predict_poses(feats, clips, sample, ds, dev)
            (pose["pred"].square().sum() + pose["conf"].sum()).backward()
            for p in model.parameters():
                p.grad = None
            feats.grad = None
        return run

    def stock(on):
        model.encoder_traj.force_stock_torch = model.encoder_traj_2d.force_stock_torch = bool(on)

    share = joint_stock_share()
    for name, feats, workload in (
            ("joint_step", None, "BASELINE configs[4] step at the reference-native grids: FORGE joint 2D3D fine-tune (predicted poses), 1 scene x 5 input + 5 novel "
             "views 256^2 -> 10 rendered views, compute_all_loss_nvs, backward incl. the pose chain, clip 10, Adam; train-mode BatchNorm / Dropout; eager launch"),
            ("joint_step_grid64", f64, "BASELINE configs[4] at its 128^3-voxel grid: the same step with 5 synthetic [128,64^3] feature volumes entering rotate(D=64) -> "
             "fusion at M=262144 -> heads -> 128^3 x 17 volume -> 10 rendered views; pose networks on their native inputs; eager launch")):
        try:
            step = make_step(feats)
            step()                                                    # allocator / MIOpen solver warm-up outside the meters
            torch.cuda.synchronize()
            from torch.utils.flop_counter import FlopCounterMode
            with FlopMeter() as fm, FlopCounterMode(display=False) as fc:
                step()
            torch.cuda.synchronize()
            ms = _timed(step, steps, warm=1)
            ms_pose = _timed(pose_nets_only(), steps, warm=1)
            e = {"name": name, "workload": workload, "steps": steps, "ms_per_step": ms, "views_per_s": 10 / ms * 1e3,
                 "roofline": dict(floor_of(fm.gflop, ms), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm.gflop / ms, launches=fm.launches,
                                  note="executed_gflop = libforge matrix-core launches (the pose estimators' convolutions included since round 5); the attention "
                                       "blocks' rocBLAS GEMMs are in stock_torch.gflop"),
                 "pose_networks": {"what": "2-D + 3-D pose estimators and pose head alone, forward + backward on the step's inputs (convolutions + BatchNorm on libforge, "
                                           "attention blocks on rocBLAS / ATen)", "fwd_bwd_ms": ms_pose, "share_of_step": ms_pose / ms},
                 "stock_torch": {"what": "kernels that are not libforge's (rocBLAS attention GEMMs, ATen element-wise / softmax / LayerNorm / optimizer): FLOPs "
                                         "counted by torch's FlopCounterMode; share of kernel time by kernel NAME from the committed rocprofv3 trace",
                                 "gflop": fc.get_total_flops() / 1e9, "rocprofv3": (share or {}).get(name)}}
            # the same step as ONE hipGraph (forge_amd.graph.GraphedStep: forward, loss, backward, clip, capturable Adam): the eager step is host-bound
            # (~3000 launches from Python); reported beside the eager number, which is what a DDP wrapper runs
            try:
                from forge_amd.graph import GraphedStep
                opt_g = torch.optim.Adam(params, lr=1e-4, capturable=True)
                call_g = model if feats is None else (lambda s, d, dv: model(s, d, dv, features_recon=feats))

                def graph_fn():
                    loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, sample, ds, call_g, {}, dev)
                    loss.backward()
                    train.clip_grad_norm_(model.parameters(), 10.0)
                    opt_g.step()
                    return loss.detach()
                gs = GraphedStep(graph_fn, opt_g, warmup=2)
                msg = _timed(gs, steps, warm=1)
                e["hipgraph_replay"] = dict(floor_of(fm.gflop, msg), ms_per_step=msg, views_per_s=10 / msg * 1e3)
                del gs, opt_g
            except Exception as ex:
                e["hipgraph_replay"] = {"error": repr(ex)[:300]}
            for p_ in model.parameters():
                p_.grad = None
            torch.cuda.empty_cache()
            if feats is None:
                # the round-4 state for comparison: the same step with both pose estimators on stock torch kernels (MIOpen picks its solvers per process:
                # asm Winograd in one, `naive_conv_*` fp32 in the next - 216 ms of a 256 ms step in BENCH-style runs of round 5's first build)
                try:
                    stock(True)
                    step()
                    torch.cuda.synchronize()
                    e["pose_networks_on_stock_torch"] = {"ms_per_step": _timed(step, steps, warm=1), "pose_nets_fwd_bwd_ms": _timed(pose_nets_only(), steps, warm=1)}
                except Exception as ex:
                    e["pose_networks_on_stock_torch"] = {"error": repr(ex)[:200]}
                finally:
                    stock(False)
            out.append(e)
        except Exception as e:
            out.append({"name": name, "workload": workload, "error": repr(e)[:300]})
        torch.cuda.empty_cache()
    # the reference's joint configuration trains 4 scenes per GPU (config/kubric/joint_pose_2d3d.yaml: batch_size 4): the GPU-bound regime of the same step
    try:
        s4 = {k: v.to(dev) for k, v in syn.make_sample(4, 10, 256, 1.5, seed=13).items()}
        step4 = make_step(None, s4)
        step4()
        torch.cuda.synchronize()
        with FlopMeter() as fm4:
            step4()
        torch.cuda.synchronize()
        ms4 = _timed(step4, max(2, steps // 2), warm=1)
        out.append({"name": "joint_step_4_scenes", "workload": "the joint step at the reference configuration's per-GPU batch (4 scenes x (5 + 5) views -> 40 rendered views per step); eager launch",
                    "steps": max(2, steps // 2), "ms_per_step": ms4, "views_per_s": 40 / ms4 * 1e3, "stock_torch": {"rocprofv3": (share or {}).get("joint_step_4_scenes")},
                    "roofline": dict(floor_of(fm4.gflop, ms4), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm4.gflop / ms4, launches=fm4.launches)})
        del s4
    except Exception as e:
        out.append({"name": "joint_step_4_scenes", "error": repr(e)[:300]})
    for p_ in model.parameters():
        p_.grad = None
    torch.cuda.empty_cache()
    return out


def _bracketed(step, steps, warm, dev):
    """`warm` untimed + `steps` timed calls of `step` between (barrier, synchronize) pairs; returns seconds per step, max over ranks."""
    sync = torch.cuda.synchronize if torch.device(dev).type == "cuda" else (lambda: None)      # the CPU / gloo rehearsal (--dry-run) has no device to wait for
    for _ in range(warm):
        step()
    fdist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    fdist.barrier()
    return fdist.all_reduce_scalars([(time.perf_counter() - t0) / steps], dev, "max")[0]


def ddp_train_record(rank, world, dev, steps, scenes=4, grid=32):
    """BASELINE configs[3] inside the driver's `--gpus N` line: the iteration of scripts/kubric_trainer.py:47-59 as
    kubric_train_pose_3D.py:119-130 wraps the model - FORGE_poseEstimator3D (GT poses) under SyncBatchNorm (HIP kernels + one all-reduce of
    2C+1 float64 per layer and direction) and DistributedDataParallel (bucketed gradient all-reduce overlapped with the backward), `scenes`
    scenes per GPU, loss, clip 10, Adam - `steps` timed steps, and the SAME step under `no_sync()` (no gradient all-reduce; SyncBatchNorm still
    exchanges its statistics), so that the all-reduce's exposed cost is a difference of two measured numbers."""
    from forge_amd import train
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
    cfg = syn.kubric_config()
    model = FORGE_poseEstimator3D(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).train()
    n_bn = sum(1 for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
    bn_ch = sum(m.num_features for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True)     # kubric_train_pose_3D.py:124
    opt = torch.optim.Adam([p for p in ddp.parameters() if p.requires_grad], lr=cfg.train.lr, fused=True)
    sample = {k: v.to(dev) for k, v in syn.make_sample(scenes, T_IN, 256, 1.5, seed=3000 + rank).items()}
    if grid == 64:
        gen = torch.Generator(device=dev).manual_seed(80 + rank)
        sample["features_recon"] = torch.randn(scenes, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    ds = syn.SyntheticDataset(1.5)
    loss = [None]

    def step():
        loss[0] = train.train_step(cfg, sample, ds, ddp, opt, dev)[0]

    def step_nosync():
        with ddp.no_sync():
            loss[0] = train.train_step(cfg, sample, ds, ddp, opt, dev)[0]
    s_sync = _bracketed(step, steps, 2, dev)
    l_sync = float(loss[0])
    s_nosync = _bracketed(step_nosync, steps, 1, dev)
    grad_bytes = sum(p.numel() for p in ddp.parameters() if p.requires_grad) * 4
    return {"workload": "BASELINE configs[3] step: FORGE_poseEstimator3D GT-pose training, %d scene(s)/GPU x 5 views -> 3 fusions -> 10 rendered views/scene, "
                        "%s, SyncBatchNorm + DDP, clip 10, Adam" % (scenes, "reference-native 32^3 / 64^3 grids" if grid == 32 else
                                                                    "128^3-voxel render grid from synthetic [128,64^3] feature volumes (encoder not run)"),
            "scenes_per_gpu": scenes, "global_batch": scenes * world, "feature_grid": grid, "steps": steps,
            "ms_per_step": s_sync * 1e3, "views_per_s": scenes * 10 * world / s_sync, "ms_per_step_no_sync": s_nosync * 1e3,
            "gradient_all_reduce_exposed_ms": (s_sync - s_nosync) * 1e3,
            "gradient_bytes_all_reduced_per_step": grad_bytes, "syncbn_layers": n_bn,
            "syncbn_bytes_all_reduced_per_step": (2 * bn_ch + n_bn) * 8 + 2 * bn_ch * 8,
            "mean_loss_all_ranks": fdist.all_reduce_scalars([l_sync], dev, "sum")[0] / world,
            "note": "no_sync = the same step without DDP's gradient all-reduce (SyncBatchNorm statistics still exchanged): the difference is the all-reduce time "
                    "the backward does not hide"}


def ray_sharded_joint_record(rank, world, dev, steps, grid=32):
    """BASELINE configs[4] inside the driver's `--gpus N` line: the joint 2D3D fine-tune iteration (kubric_train_joint.py:136-141 -> compute_all_loss_nvs)
    with the ray-march of its 10 views split into row bands over the N ranks (train.enable_ray_sharding: all_gather of the rendered maps forward,
    all-reduce of d(volume) / d(cameras) backward; encoder / pose networks / fusion / conv_rgb replicated on the SAME batch, DDP keeps the replicas
    identical), next to the same step unsharded on every rank; plus the sharded render op alone, forward + backward, in both reduce modes (all-reduce /
    reduce-to-owner through dist.broadcast_from_owner) on the 64^3 and the 128^3 volume."""
    from forge_amd import ops, train
    from forge_amd.model import FORGE
    cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
    cfg.loss.regu_origin_proj = 1.0
    torch.manual_seed(1234)                                            # every rank draws the same Dropout masks: the replicas must predict the same poses
    model = FORGE(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model.to(dev).train())                                        # kubric_train_joint.py:136 (HIP SyncBatchNorm: one all-reduce per layer and direction)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True)         # kubric_train_joint.py:141
    params = [p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head, model.render) for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}           # the same scene on every rank (train_step broadcasts rank 0's anyway)
    if grid == 64:
        gen = torch.Generator(device=dev).manual_seed(79)
        sample["features_recon"] = torch.randn(1, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    ds = syn.SyntheticDataset(1.5)
    loss = [None]

    def step():
        loss[0] = train.train_step(cfg, sample, ds, ddp, opt, dev, loss_func=train.compute_all_loss_nvs)[0]
    out = {"workload": "BASELINE configs[4] step: FORGE joint 2D3D fine-tune (predicted poses), 1 scene x 5 input + 5 novel views -> 10 rendered views, rays of every "
                       "view sharded over the ranks in row bands; %s; DDP over the replicas" % ("reference-native grids" if grid == 32 else "128^3-voxel render grid (synthetic 64^3 features)"),
           "feature_grid": grid, "steps": steps, "band_rows": 128 // world if 128 % world == 0 else None}
    train.enable_ray_sharding(ddp, False)
    s_full = _bracketed(step, steps, 2, dev)
    out["unsharded_ms_per_step"] = s_full * 1e3
    if 128 % world == 0:
        train.enable_ray_sharding(ddp, True, reduce="all")
        s_shard = _bracketed(step, steps, 1, dev)
        out.update(ms_per_step=s_shard * 1e3, views_per_s=10 / s_shard, loss=float(loss[0]))
        train.enable_ray_sharding(ddp, False)
    Dr = 2 * grid
    out["all_gather_bytes_per_step"] = 10 * 17 * 128 * 128 * 4
    out["all_reduce_bytes_per_step"] = 17 * Dr ** 3 * 4 + 10 * 16 * 4
    del ddp, opt
    # the sharded render op alone: forward (all_gather) + backward, reduce "all" (all-reduce of d volume) vs "none" + broadcast_from_owner (reduce to the owner)
    op = {}
    if 128 % world == 0 and grid == 32:                               # once per line (the grid-64 record does not repeat it)
        _, extr, _ = syn.orbit_cameras(10, 1.5, 10.0)
        K = syn.intrinsics(256) / 2.0
        cam = torch.cat([extr[:, :3, :3].reshape(10, 9), extr[:, :3, 3], K[0, 0].expand(10, 1), K[1, 1].expand(10, 1), K[0, 2].expand(10, 1), K[1, 2].expand(10, 1)],
                        dim=1).contiguous().to(dev)
        v2v = torch.zeros(10, dtype=torch.int32, device=dev)
        for D in (64, 128):
            feat0, dens0 = syn.blob_volumes(1, D, 16, seed=0)
            feat = feat0.to(dev).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3).requires_grad_(True)
            dens = dens0.to(dev).contiguous().requires_grad_(True)
            h = 0.5 * (D - 1) / D
            for mode in ("all", "none"):
                def run():
                    f, d = (feat, dens) if mode == "all" else fdist.broadcast_from_owner((feat, dens), src=0)
                    o = fdist.render_rays_sharded(f, d, cam, v2v, 128, 128, 64, 0.5, 2.0, (h, h, h), reduce=mode)
                    (o[0].square().sum() + o[1].sum()).backward()
                    feat.grad = dens.grad = None
                op["volume_%d_reduce_%s_fwd_bwd_ms" % (D, mode)] = _bracketed(run, max(3, steps), 1, dev) * 1e3
            op["volume_%d_bytes" % D] = 17 * D ** 3 * 4
    if op:
        out["sharded_render_op"] = op
    return out


def multi_rank_records(args, rank, world, dev, minimal, records=None):
    """world > 1 only, all ranks, BEFORE the process group is torn down: the two sub-records whose collectives matter on an 8-GPU node (DDP +
    SyncBatchNorm training, the ray-sharded joint step) - bounded (<= 5 steps each), each in its own try block, under a watchdog: if the records do
    not finish within the deadline (a rank that died inside a collective leaves the others waiting), rank 0 prints the MAIN line with what it has
    (`minimal`) and every rank leaves - the driver's line never depends on the sub-records."""
    import threading
    deadline = float(os.environ.get("FORGE_BENCH_SUBRECORD_DEADLINE_S", "420"))
    done = threading.Event()

    def watchdog():
        if done.wait(deadline + (0 if rank == 0 else 10)):
            return
        if rank == 0:
            print(json.dumps(dict(minimal, multi_rank={"error": "sub-records did not finish within %.0f s; main line printed by the watchdog" % deadline})), flush=True)
        os._exit(0)
    threading.Thread(target=watchdog, daemon=True).start()
    rec = {}
    n = max(1, min(5, args.steps))
    if records is None:
        records = (("ddp_train", lambda: ddp_train_record(rank, world, dev, n, scenes=4, grid=32)),
                   ("ray_sharded_joint", lambda: ray_sharded_joint_record(rank, world, dev, n, grid=32)),
                   ("ray_sharded_joint_grid64", lambda: ray_sharded_joint_record(rank, world, dev, max(1, min(3, n)), grid=64)))
    cuda = torch.device(dev).type == "cuda"
    for name, fn in records:
        err = None
        try:
            r = fn()
        except Exception as e:                                       # reported per rank; a failure INSIDE a collective is what the watchdog is for
            import traceback
            traceback.print_exc()
            r, err = None, repr(e)[:300]
        if cuda:
            torch.cuda.empty_cache()
        errs = [e for e in fdist.gather_strings(err) if e]
        ok = fdist.all_reduce_scalars([0.0 if err else 1.0], dev, "sum")[0]
        rec[name] = dict(r or {}, ranks_ok=int(ok), errors=errs, process_group=fdist.group_info())
    done.set()
    return rec


def train_bench(args, rank, world, dev, affinity):
    """`--train`: BASELINE configs[3] as a scaling measurement - FORGE_poseEstimator3D (GT poses), args.scenes scenes per GPU x 5 views ->
    10 rendered views per scene, SyncBatchNorm (HIP kernels, one RCCL all-reduce of the float64 statistics per layer and direction) +
    DistributedDataParallel (bucketed RCCL gradient all-reduce overlapped with the backward), loss, clip 10, Adam: the iteration of
    scripts/kubric_trainer.py:47-59 as kubric_train_pose_3D.py:119-124 wraps the model. Prints its own metric string."""
    from forge_amd import train
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
    cfg = syn.kubric_config()
    B = args.scenes
    ok, err, dt, loss = 1.0, None, 0.0, float("nan")
    fdist.init(allow_shared_gpus=os.environ.get("FORGE_BENCH_ALLOW_SHARED_GPUS") == "1")
    try:
        model = FORGE_poseEstimator3D(cfg)
        model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
        model = model.to(dev).train()
        if world > 1:
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True)     # kubric_train_pose_3D.py:124
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg.train.lr, fused=True)
        sample = {k: v.to(dev) for k, v in syn.make_sample(B, T_IN, 256, 1.5, seed=1000 + rank).items()}
        if args.grid == 64:
            # configs[3]'s REAL per-GPU shape: the 128^3-voxel render grid = 64^3 feature grid; synthetic feature volumes ride in the sample
            # (FORGE_poseEstimator3D.forward(features_recon=): the encoder cannot produce them from 256^2 images and is not run)
            gen = torch.Generator(device=dev).manual_seed(78 + rank)
            sample["features_recon"] = torch.randn(B, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
        ds = syn.SyntheticDataset(1.5)

        def step():
            return train.train_step(cfg, sample, ds, model, opt, dev)[0]
    except Exception as e:
        ok, err = 0.0, repr(e)[:400]
        import traceback
        traceback.print_exc()
    R = max(1, min(args.repeats, 3))                                # bounded: a training region is steps x ~0.2 s
    dts = [0.0] * R
    if ok:
        ok, err, lt, dts = timed_region(step, args.steps, args.warmup, R)
        loss = float(lt) if ok else float("nan")
    else:
        for _ in range(2 * R):
            fdist.barrier()
    pg = fdist.group_info()
    dts = fdist.all_reduce_scalars(dts, dev, "max")
    dt = region_stats(dts, args.steps, 1.0)[0]                      # median region
    views, ranks_ok, loss_sum = fdist.all_reduce_scalars([B * 10.0 * ok, ok, loss if ok else 0.0], dev, "sum")
    errs = fdist.gather_strings(err)
    if rank == 0:
        ms = dt / args.steps * 1e3 if dt > 0 else None
        print(json.dumps({
            "metric": "rendered views/sec incl. backward (GT-pose training step, 10 views/scene, %d^3 voxel)" % (2 * args.grid), "value": views * args.steps / dt if dt > 0 else None,
            "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_ok": int(ranks_ok), "errors": [e for e in errs if e],
            "process_group": pg, "repeats": region_stats(dts, args.steps, views)[1] if dt > 0 else None,
            "mean_loss_all_ranks": loss_sum / max(ranks_ok, 1.0),
            "config": {"workload": "BASELINE configs[3] step: FORGE_poseEstimator3D GT-pose training, %d scene(s)/GPU x 5 views -> 3 fusions -> 10 rendered "
                                   "views/scene, %s, SyncBatchNorm + DDP" % (B, "reference-native 32^3 / 64^3 grids" if args.grid == 32 else "128^3-voxel render grid from synthetic "
                                   "[128,64^3] feature volumes (encoder not run)"), "scenes_per_gpu": B, "feature_grid": args.grid,
                       "global_batch": B * world, "parallelism": "dp%d (DDP bucketed RCCL all-reduce of 221 MB fp32 gradients; HIP SyncBatchNorm)" % world,
                       "rank0_affinity": affinity}}), flush=True)
    fdist.barrier()
    fdist.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenes", type=int, default=1, help="scenes per GPU per step (BASELINE configs[1]: 1, configs[2]: 8)")
    ap.add_argument("--grid", type=int, default=32, choices=(32, 64),
                    help="feature grid: 32 = the metric's configuration (64^3 render volume); 64 = BASELINE configs[3]/[4] 128^3-voxel "
                         "scenes: synthetic [b,5,128,64^3] feature volumes through rotate -> fuse -> heads -> ray-march (the encoder cannot produce them)")
    ap.add_argument("--train", action="store_true", help="time the data-parallel training step (SyncBatchNorm + DDP) instead of inference")
    ap.add_argument("--repeats", type=int, default=10,
                    help="timed regions of EXACTLY --steps steps each in the same run (each between barrier + synchronize pairs); value = units / the MEDIAN "
                         "region, min / max reported beside it")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--pipeline-depth", type=int, default=4,
                    help="steps in flight: that many hipGraphs of the step replayed round-robin on as many HIP streams (forge_amd.graph.PipelinedForward); "
                         "1 = one stream, back to back")
    ap.add_argument("--dump-conv", action="store_true", help="print every conv launch of one step (shape, ms, TFLOP/s) to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="with --no-cpu-baseline: also skip the one CPU oracle forward the last output is checked against")
    ap.add_argument("--min-psnr-db", type=float, default=None, help="exit non-zero unless the last output of the timed region is at least this close to the oracle "
                                                                    "(soak runs: --steps 3000 --no-cpu-baseline --no-extra --min-psnr-db 100)")
    ap.add_argument("--no-microbench", action="store_true", help="skip the per-kernel micro-benchmarks (clean rocprofv3 stats)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs / strong_scaling (the other BASELINE configurations)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the multi-rank launch path (no HIP work)")
    ap.add_argument("--rehearse-hang", action="store_true", help=argparse.SUPPRESS)       # --dry-run only: one sub-record blocks for ever (the watchdog's test)
    ap.add_argument("--cpu-worker", nargs=3, type=int, metavar=("THREADS", "N", "SEED"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(*args.cpu_worker)
    self_launch(args)

    rank, local_rank, world = fdist.env_world()
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launch environment has WORLD_SIZE=%d (run `python bench.py --gpus N` and let it "
                         "start its own ranks, or pass matching values to torch.distributed.run)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    ndev = torch.cuda.device_count()
    if world > ndev and os.environ.get("FORGE_BENCH_ALLOW_SHARED_GPUS") != "1":
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible - a scaling number from shared devices would be meaningless "
                         "(set FORGE_BENCH_ALLOW_SHARED_GPUS=1 for a functional rehearsal)" % (world, ndev))
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    affinity = pin_rank_to_gpu_numa(dev) if world > 1 else {"pinned": False, "reason": "single rank"}
    _lib.lib()
    if args.train:
        return train_bench(args, rank, world, dev, affinity)

    from forge_amd.model import FORGE
    cfg = syn.kubric_config()
    B = args.scenes
    ok, err = 1.0, None
    graphed = step = eager_step = strong = None
    B_strong = max(1, 8 // world) if (world > 1 and not args.no_extra and args.grid == 32 and not args.no_graph) else 0
    try:
        model = FORGE(cfg)
        weights = syn.seeded_state_dict(model.state_dict(), 0)
        model.load_state_dict(weights)
        model = model.to(dev).eval()
        sample_cpu = syn.make_sample(B, T_IN, 256, 1.5, seed=1000 + rank)
        sample = {k: v.to(dev) for k, v in sample_cpu.items()}      # inputs resident in HBM
        dataset = syn.SyntheticDataset(1.5)

        if args.grid == 32:
            def eager_step():
                with torch.no_grad():
                    return model(sample, dataset, dev)
        else:
            # 128^3-voxel scenes: per-view feature volumes [B,5,128,64^3] (671 MB per scene) resident in HBM, GT poses / cameras of the sample
            from forge_amd import geo_utils
            gen = torch.Generator(device=dev).manual_seed(77 + rank)
            feats64 = torch.randn(B, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
            poses64 = sample["cam_poses_cv2_canonicalized"][:, :T_IN].contiguous()
            cams64 = geo_utils.camera_dict(sample["cam_extrinsics_cv2_canonicalized"][:, :V_OUT], sample["K_cv2"][:, :V_OUT])

            def eager_step():
                with torch.no_grad():
                    return model.reconstruct(feats64, poses64, cams64)[:2]

        # hipGraph capture happens BEFORE the process group exists: no RCCL communicator / watchdog thread is alive while the stream is
        # capturing, so the capture cannot be invalidated by collective-library activity; the barrier / all-reduce below never run inside it.
        if args.no_graph:
            step = eager_step
        elif args.grid == 32:
            # hipGraph(s) of the whole step; replays do all the work. pipeline_depth steps are kept in flight on as many HIP streams: the
            # under-filled ResNet launches of one step share the chip with the MFMA-bound ConvGRU launches of its neighbours
            from forge_amd.graph import PipelinedForward
            graphed = PipelinedForward(model, sample, dataset, dev, depth=max(1, args.pipeline_depth))
            step = lambda: graphed(sample)                              # noqa: E731  (copies the resident inputs into the slot's static buffers)
        else:
            from forge_amd.graph import GraphedCall
            step = GraphedCall(eager_step, dev)
        if B_strong and B_strong != B:                                   # strong scaling: 8 scenes in total over the N ranks
            from forge_amd.graph import PipelinedForward
            s_strong = {k: v.to(dev) for k, v in syn.make_sample(B_strong, T_IN, 256, 1.5, seed=2000 + rank).items()}
            g_strong = PipelinedForward(model, s_strong, dataset, dev, depth=2)
            strong = lambda: g_strong(s_strong)                          # noqa: E731
        elif B_strong:
            strong = step
    except Exception as e:                                                # this rank still joins the rendezvous and the reductions: no hang
        ok, err = 0.0, repr(e)[:400]
        import traceback
        traceback.print_exc()

    fdist.init(allow_shared_gpus=os.environ.get("FORGE_BENCH_ALLOW_SHARED_GPUS") == "1")      # RCCL (backend "nccl") over xGMI when world > 1
    fdist.barrier()

    R = max(1, args.repeats)
    if ok:
        ok, err, out, dts = timed_region(step, args.steps, args.warmup, R)
    else:
        for _ in range(2 * R):
            fdist.barrier()
        out, dts = None, [0.0] * R
    pg = fdist.group_info()                                          # which backend actually carried the collectives of this run
    dts = fdist.all_reduce_scalars(dts, dev, "max")                  # every region: the slowest rank's clock
    dt = region_stats(dts, args.steps, 1.0)[0]                      # median region
    # the one exchange of the inference path: (SSE to the target views, pixel count, views rendered) summed over ranks
    # (RCCL all-reduce of a few doubles) -> whole-job PSNR / view count; ranks_ok rides along
    if ok:
        tgt_dev = sample["images"][:, :V_OUT].reshape(B * V_OUT, 3, 256, 256)
        sse_local, npix_local = float(((out[0] - tgt_dev) ** 2).sum()), float(tgt_dev.numel())
    else:
        sse_local = npix_local = 0.0
    sse, npix, views_per_step, ranks_ok = fdist.all_reduce_scalars([sse_local, npix_local, float(B * V_OUT) * ok, ok], dev, "sum")
    errors = [e for e in fdist.gather_strings(err) if e]
    views = int(views_per_step) * args.steps

    strong_res = None
    if B_strong:                                                     # bounded: <= 5 steps
        n_s = min(5, args.steps)
        if strong is not None and ok:
            ok_s, err_s, _, dts_s = timed_region(strong, n_s, 1)
            dt_s = dts_s[0]
        else:
            fdist.barrier()
            fdist.barrier()
            ok_s, dt_s = 0.0, 0.0
        dt_s = fdist.all_reduce_scalars([dt_s], dev, "max")[0]
        v_s, r_s = fdist.all_reduce_scalars([float(B_strong * V_OUT) * ok_s, ok_s], dev, "sum")
        strong_res = {"scaling": "strong", "total_scenes": B_strong * world, "scenes_per_gpu": B_strong, "steps": n_s, "ms_per_step": dt_s / n_s * 1e3,
                      "views_per_s": v_s * n_s / dt_s if dt_s > 0 else None, "ranks_ok": int(r_s),
                      "note": "8 scenes in total split over the ranks; the N = 1 point of this curve is extra_configs['configs[2]'] of the --gpus 1 line"}

    # world > 1: the sub-records whose collectives matter (DDP + SyncBatchNorm training, ray-sharded joint step), bounded and under a watchdog that
    # prints the main line below if they do not come back
    multi = None
    if world > 1 and not args.no_extra and args.grid == 32:
        metric_main = "rendered views/sec (5 views, 128^2 px, 64^3 voxel)"
        minimal = {"metric": metric_main, "value": (int(views_per_step) * args.steps / dt) if (ok and dt > 0) else None, "unit": "views/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_ok": int(ranks_ok), "errors": errors, "process_group": pg,
                   "strong_scaling": strong_res, "config": {"workload": "BASELINE configs[1]: FORGE hot path, %d scene(s)/GPU x 5 views (see the full line of a run "
                                                                        "whose sub-records finished)" % B, "scenes_per_gpu": B}}
        if graphed is not None:
            graphed.wait()
        multi = multi_rank_records(args, rank, world, dev, minimal)
    # every rank is done with collectives: tear the process group down NOW, so that rank 0's per-kernel measurements, the other
    # configurations and the CPU baseline below never keep the other ranks (or an RCCL watchdog) waiting
    fdist.barrier()
    fdist.shutdown()
    if rank != 0:
        return None
    if not ok:
        print(json.dumps({"metric": "rendered views/sec (5 views, 128^2 px, 64^3 voxel)", "value": None, "unit": "views/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ranks_ok": int(ranks_ok), "process_group": pg, "errors": errors, "error": err}), flush=True)
        return None

    # ---- the same steps with the sample handed over as (pinned) HOST buffers, as a DataLoader would: PCIe-inclusive rate (never `value`)
    pcie_views_per_s = None
    if world == 1 and graphed is not None:
        graphed.wait()
        host = {k: v.pin_memory() for k, v in sample_cpu.items()}
        graphed(host)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            graphed(host)
        torch.cuda.synchronize()
        pcie_views_per_s = B * V_OUT * args.steps / (time.perf_counter() - t1)

    # ---- the same replay on ONE stream, back to back (= the latency of a step), and the per-stage split of that replay
    single = stages_replay = None
    if graphed is not None:
        one = graphed.slots[0]
        ms1 = _timed(lambda: one(sample), max(5, min(20, args.steps)))
        single = {"ms_per_step": ms1, "views_per_s": B * V_OUT / ms1 * 1e3, "note": "one hipGraph replay at a time on one stream: step latency"}
        if args.grid == 32 and not args.no_microbench:
            from forge_amd.flopmeter import stage_replay_ms
            stages_replay = {k: round(v, 4) for k, v in stage_replay_ms(model, sample, dev).items()}
    # ---- per-stage HIP-event split of one more step (outside the timed region)
    rec, undo = stage_timers(model)
    for _ in range(3):
        rec.clear()
        eager_step()
    torch.cuda.synchronize()
    conv_rec = {k: rec.pop(k) for k in list(rec) if k.startswith("conv_igemm")}
    wino_rec = {k: rec.pop(k) for k in list(rec) if k.startswith("wino_")}
    stages = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in rec.items()}
    # x[4] (Winograd point-GEMM launches only): direct-convolution FLOPs of the convolution / FLOPs the launch executes
    conv_launch = {k: {"launches_per_step": len(v), "total_ms": sum(x[0].elapsed_time(x[1]) for x in v),
                       "gflop": sum(x[2] for x in v) / 1e9, "gflop_direct_equivalent": sum(x[2] * (x[4] if len(x) > 4 else 1.0) for x in v) / 1e9}
                   for k, v in conv_rec.items()}
    for u in undo:
        u()
    if args.dump_conv:
        for k, v in conv_rec.items():
            for x in v:
                ms = x[0].elapsed_time(x[1])
                print("%-34s M=%-7d N=%-5d taps=%-3d Cin=%-5d %.4f ms  %.1f TF" % ((k,) + x[3] + (ms, x[2] / ms / 1e9)), file=sys.stderr)
    if "encoder_total" in stages:
        stages["encoder_conv1(+layout)"] = stages.pop("encoder_total") - stages.get("encoder_resnet", 0.0)
    stages["render_march(+cam pack)"] = stages.pop("render_total") - stages.get("conv_rgb", 0.0)

    kern = {} if args.no_microbench else kernel_rooflines(dev, B, args.grid)
    for k, v in wino_rec.items():          # Winograd transform kernels of the fusion, as launched inside the step
        ms, by = sum(x[0].elapsed_time(x[1]) for x in v), sum(x[2] for x in v)
        kern[k] = {"bound": "hbm", "launches_per_step": len(v), "ms_total": ms, "bytes": by, "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "traffic": pmc_traffic(k),
                   "note": "HIP events around the eager launches of one step (each includes the host launch gap); the transformed operands "
                           "(67-134 MB per launch at one scene) are partly served by the 256 MB Infinity Cache"}
    # dominant kernel of the step: conv_igemm_kernel<BM, BN, waves> - ONE kernel (csrc/conv_igemm.hip) whose tile shape is picked per
    # launch by the plan model, so rocprofv3 lists it under several instantiation names; together they are ~85 % of the step.
    # achieved = sum of the FLOPs its launches EXECUTE in one step (direct convolutions 2 M N taps Cin, Winograd point-GEMM launches
    # 2 x 16 R N kd Cin) / sum of their HIP-event durations. The per-instantiation avg_launch_ms are directly comparable with
    # rocprofv3's per-name AverageNs in profiles/. floor_ms = the same executed FLOPs at the 157.3 TF pipe peak: the step's own time floor.
    convs = {k: v for k, v in conv_launch.items() if k.startswith("conv_igemm_kernel<")}
    step_ms = dt / args.steps * 1e3
    tot_ms = sum(v["total_ms"] for v in convs.values())
    tot_gf = sum(v["gflop"] for v in convs.values())
    n_launch = sum(v["launches_per_step"] for v in convs.values())
    alg_gf = sum(v["gflop_direct_equivalent"] for v in convs.values())
    n16_gf = sum(v["gflop"] for k, v in conv_launch.items() if not k.startswith("conv_igemm_kernel<"))
    inst = {k: {"launches_per_step": v["launches_per_step"], "avg_launch_ms": v["total_ms"] / v["launches_per_step"],
                "achieved": v["gflop"] / v["total_ms"], "frac": v["gflop"] / v["total_ms"] / FP32_MFMA_PEAK_TF,
                "gflop_per_step": v["gflop"], "share_of_step": v["total_ms"] / step_ms}
            for k, v in sorted(convs.items(), key=lambda kv: -kv[1]["total_ms"])}
    fl = floor_of(tot_gf + n16_gf, step_ms)
    tr = pmc_traffic("winograd gates" if wino_rec else "conv_igemm_kernel<128")
    rp = rocprof_conv_time() if (args.grid == 32 and B == 1) else None
    ceil_tf = KLOOP_CEILING_TF["64x128"]                              # the tile that carries ~80 % of the step's FLOPs
    roofline = {"kernel": "conv_igemm_kernel<BM, BN, waves> (fp32 MFMA implicit-GEMM conv; all %d launches of one step, %d tile instantiations)"
                          % (n_launch, len(convs)),
                "bound": "mfma", "achieved": tot_gf / tot_ms, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tot_gf / tot_ms / FP32_MFMA_PEAK_TF,
                # flat keys: HBM-side bytes of the dominant launch from the PMC counters, per launch, next to its algorithmic bytes
                "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_algorithmic_bytes": tr["algorithmic_bytes"] if tr else None,
                "traffic_launch": tr["launch"] if tr else None, "traffic_source": ("profiles/" + tr["source"]) if tr else None,
                # (5b) the same fraction from pure kernel durations: rocprofv3 --kernel-trace --stats of this command, one step in flight
                "frac_rocprof": (tot_gf / rp["ms_per_step"] / FP32_MFMA_PEAK_TF) if rp else None, "rocprof_kernel_ms_per_step": rp["ms_per_step"] if rp else None,
                "rocprof_source": ("profiles/" + rp["source"]) if rp else None,
                # (5c) what the kernel's own LDS -> MFMA loop can do with staging removed (measured on debug builds): the exact-fp32 ceiling of this design
                "ceiling": {"kloop_without_staging_tflops": KLOOP_CEILING_TF, "frac_of_peak": ceil_tf / FP32_MFMA_PEAK_TF,
                            "source": "profiles/TUNING_LOG.md 'K-loop ceiling' (tools/debug/gemm_ceiling.py on FORGE_EXP_* debug builds, direct gates launch K = 6912)"},
                "frac_of_ceiling": (tot_gf / (rp["ms_per_step"] if rp else tot_ms)) / ceil_tf,
                "avg_launch_ms": tot_ms / n_launch,
                "executed_gflop": fl["executed_gflop"], "executed_frac": fl["executed_frac"], "floor_ms": fl["floor_ms"], "step_over_floor": fl["step_over_floor"],
                "kernel_ms_per_step": tot_ms, "share_of_step": tot_ms / step_ms, "instantiations": inst,
                "note": "frac = FLOPs the dominant kernel's launches EXECUTE / their HIP-event time / peak (a statement about the kernel; eager pass, each "
                        "event pair includes the host launch gap and, for split-K launches, the reduction); frac_rocprof = the same FLOPs / the kernels' own "
                        "durations in the committed rocprofv3 trace of this command. executed_frac = floor_ms / ms_per_step = the "
                        "WHOLE step (all kernels, hipGraph replay) against the time its executed matrix-core FLOPs need at peak (a statement about the "
                        "step). In SURVEY.md 8(d)'s direct-convolution FLOPs the same launches are %.0f GF (the Winograd launches execute 2.25x fewer "
                        "multiplies than the convolutions they replace), so a fraction in those units can exceed 1 and is not reported as one; "
                        "traffic = PMC pass of the fusion's point-GEMM launch (L2 -> fabric bytes, Infinity-Cache hits included)" % alg_gf}
    if args.grid == 32:
        metric = "rendered views/sec (5 views, 128^2 px, 64^3 voxel)"
        workload = ("BASELINE configs[%d]: FORGE hot path, %d scene(s)/GPU x 5 input views 256^2 -> 32^3x128 feature "
                    "grid -> 64^3 render grid -> 5 views x 128^2 rays x 64 samples -> 5 RGB 256^2; HIP rotate, "
                    "fp32-MFMA implicit-GEMM ResNet-50 trunk / conv1 / ConvGRU (Winograd F(2x2,3x3) x 3 depth taps) / heads / conv_rgb, HIP ray-march (no MIOpen/rocBLAS kernel in the step); "
                    "eval BN, random-init seeded weights" % (1 if B == 1 else 2, B))
        gflop = B * (GF_ENCODER + GF_FUSE + GF_HEADS + GF_CONVRGB)
    else:
        metric = "rendered views/sec (5 views, 128^2 px, 128^3 voxel)"
        workload = ("BASELINE configs[3]/[4] grid (synthetic up-scale, SURVEY.md 8d): %d scene(s)/GPU x 5 synthetic feature volumes "
                    "[128,64^3] resident in HBM (the encoder cannot produce them from 256^2 images, models/encoder.py:49) -> HIP rotate at "
                    "D=64 (1.07 GB/scene) -> ConvGRU fusion at M=262144 -> heads -> 128^3 x 17 render volume (142.6 MB) -> 5 views x "
                    "128^2 rays x 64 samples -> conv_rgb -> 5 RGB 256^2; eval BN, random-init seeded weights" % B)
        gflop = B * (8 * (GF_FUSE + GF_HEADS) + GF_CONVRGB)
    result = {
        "metric": metric, "value": views / dt, "unit": "views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ranks_ok": int(ranks_ok), "errors": errors, "process_group": pg,
        "repeats": region_stats(dts, args.steps, float(views_per_step))[1],
        "config": {"workload": workload, "scenes_per_gpu": B, "views_in": T_IN, "views_out": V_OUT, "feature_grid": args.grid,
                   "render_grid": 2 * args.grid, "rank0_affinity": affinity,
                   "steps_in_flight": graphed.depth if graphed is not None else 1,
                   "launch": "eager" if args.no_graph else ("hipGraph replay, %d steps in flight on %d HIP streams" % (graphed.depth, graphed.depth)
                                                            if (graphed is not None and graphed.depth > 1) else "hipGraph replay"),
                   "parallelism": "dp%d (scene-sharded, no data-path collective; 4-scalar RCCL all-reduce of SSE/pixels/views/ok for the PSNR report)" % world},
        "single_stream": single,
        "roofline": roofline, "conv_launches": conv_launch, "kernels": kern, "stages_ms": {k: round(v, 4) for k, v in stages.items()},
        "stages_ms_replay": stages_replay,
        "gflop_per_step_algorithmic": gflop,
        "views_per_s_with_host_to_device_copy": pcie_views_per_s,
        "psnr_to_target_db_all_ranks": fdist.psnr_from_sse(sse, npix),
    }
    if strong_res is not None:
        result["strong_scaling"] = strong_res
    if multi is not None:
        result["multi_rank"] = multi
    ref = None
    if world == 1 and args.grid == 32 and not (args.no_cpu_baseline and args.no_oracle_check):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import forge_oracle as fo
        if not args.no_cpu_baseline:
            cb, ref = cpu_baseline(sample_cpu, weights, cfg)
            result["cpu_baseline"] = cb
        else:
            # no timing of the CPU path, but the LAST output of the timed region is still checked against the oracle (one CPU forward of scene 0):
            # a soak run must look at what it produced ("a soak that never looks at its output proves only that nothing crashed")
            one = {k: v[:1] for k, v in sample_cpu.items()}
            with torch.no_grad():
                ref = fo.forward_hot_path(one["images"][:, :T_IN], one["cam_poses_cv2_canonicalized"][:, :T_IN], one["cam_extrinsics_cv2_canonicalized"][:, :T_IN],
                                          one["K_cv2"][:, :T_IN], weights, cfg, order_by_distance=True)
        img0 = out[0][:V_OUT].cpu()
        result["psnr_vs_oracle_db"] = fo.psnr(img0, ref[0])
        result["oracle_note"] = ("oracle = oracle/forge_oracle.py, pinned by golden vectors from the reference's own module code; its ray-marcher restates "
                                 "PyTorch3D 0.7.0 (not installable offline): parity with the PyTorch3D BINARY is unpinned (DESIGN.md section 4)")
        result["max_abs_err_vs_oracle"] = (img0 - ref[0]).abs().max().item()
        # north_star: "PSNR within 0.1 dB of reference" - PSNR of both against the same target images (the scene's input views; with
        # random-init weights the absolute value is meaningless, the DIFFERENCE is the criterion)
        tgt = sample_cpu["images"][0, :V_OUT]
        p_build, p_oracle = fo.psnr(img0, tgt), fo.psnr(ref[0], tgt)
        result["psnr_to_target_db"] = {"build": p_build, "oracle": p_oracle, "abs_diff": abs(p_build - p_oracle)}
        if "cpu_baseline" in result:
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
    if world == 1 and not args.no_extra and args.grid == 32:
        del graphed, step
        torch.cuda.empty_cache()
        result["extra_configs"] = extra_configs(dev, steps=10)
    print(json.dumps(result), flush=True)
    if args.min_psnr_db is not None:
        got = result.get("psnr_vs_oracle_db")
        if got is None or not got >= args.min_psnr_db:
            raise SystemExit("bench.py: the last output of the timed region is %s dB from the oracle, below --min-psnr-db %.1f" % (got, args.min_psnr_db))
    return result


if __name__ == "__main__":
    main()
