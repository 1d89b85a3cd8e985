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
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the HSA runtime starts (first torch.cuda call): this host driver supports dmabuf IPC only - a rank launched by somebody else's
# torch.distributed.run (not through benchkit.launch.rank_env) must see it too, or RCCL's cross-process buffer exchange fails
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from forge_amd import _lib, dist as fdist  # noqa: E402
from benchkit.launch import pin_rank_to_gpu_numa, self_launch  # noqa: E402


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
    ap.add_argument("--no-oracle-check", action="store_true", help="with --no-cpu-baseline: also skip the one CPU oracle forward the last output is "
            "checked against")
    ap.add_argument("--min-psnr-db", type=float, default=None, help="exit non-zero unless the last output of the timed region is at least this close to "
            "the oracle "
                                                                    "(soak runs: --steps 3000 --no-cpu-baseline --no-extra --min-psnr-db 100)")
    ap.add_argument("--no-microbench", action="store_true", help="skip the per-kernel micro-benchmarks (clean rocprofv3 stats)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs / strong_scaling (the other BASELINE configurations)")
    ap.add_argument("--full-record", default=None, metavar="PATH",
                    help="where the full record goes (default: bench_full.json beside bench.py, and gpurun_out/ when present); "
                         "stdout carries ONE compact strict-JSON line (benchkit/emit.py)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the multi-rank launch path (no HIP work)")
    ap.add_argument("--rehearse-hang", action="store_true", help=argparse.SUPPRESS)       # --dry-run only: one sub-record blocks for ever (the watchdog's test)
    ap.add_argument("--cpu-worker", nargs=3, type=int, metavar=("THREADS", "N", "SEED"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        from benchkit.cpu import cpu_worker
        return cpu_worker(*args.cpu_worker)
    self_launch(args)

    rank, local_rank, world = fdist.env_world()
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launch environment has WORLD_SIZE=%d (run `python bench.py --gpus N` and let it "
                         "start its own ranks, or pass matching values to torch.distributed.run)" % (args.gpus, world))
    if args.dry_run:
        from benchkit.dryrun import dry_run
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
        from benchkit.trainmode import train_bench
        return train_bench(args, rank, world, dev, affinity)
    from benchkit.headline import run_headline
    return run_headline(args, rank, world, dev, affinity)


if __name__ == "__main__":
    main()
