"""Constants (MI355X peaks, SURVEY.md 8d work figures) and the timing brackets every bench module shares."""
import os
import time

import torch

from forge_amd import dist as fdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32 dense peak
T_IN, V_OUT = 5, 5
# algorithmic work per scene
GF_ENCODER = 64.3 * T_IN
GF_FUSE = 927.7
GF_HEADS = 45.3
GF_CONVRGB = 0.80 * V_OUT
# what every training-step number of this bench leaves out / cannot promise
TRAIN_CAVEATS = ("; perceptual term excluded (config/kubric/gt_pose.yaml:37 perceptual_img 0.02 needs VGG-16 weights: SURVEY.md 2 row 8, out of scope); "
                 "weight gradients accumulate with fp32 atomics (csrc/conv_wgrad.hip): not bit-reproducible run to run")
# LDS -> MFMA loop alone (no global -> LDS staging), direct gates launch K = 6912: debug builds of tools/debug/gemm_ceiling.py,
# profiles/TUNING_LOG.md "K-loop ceiling"
KLOOP_CEILING_TF = {"64x64": 130.0, "64x128": 137.0, "128x128": 141.0}


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


def floor_of(gflop, ms):
    """A step against its own executed-FLOP time floor on the fp32 MFMA pipe."""
    floor_ms = gflop / FP32_MFMA_PEAK_TF
    return {"executed_gflop": gflop, "floor_ms": floor_ms, "executed_frac": floor_ms / ms if ms > 0 else None,
            "step_over_floor": ms / floor_ms if floor_ms > 0 else None}


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
