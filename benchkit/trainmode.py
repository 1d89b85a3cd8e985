"""`--train`: the data-parallel training step as its own scaling measurement (never the driver's line)."""
import os

import torch

from forge_amd import dist as fdist, synthetic as syn
from benchkit.common import T_IN, TRAIN_CAVEATS, region_stats, timed_region
from benchkit.emit import emit


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
            sample["features_recon"] = torch.randn(B, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5,
                    2).contiguous().permute(0, 1, 5, 2, 3, 4)
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
        emit({
            "metric": "rendered views/sec incl. backward (GT-pose training step, 10 views/scene, %d^3 voxel)" % (2 * args.grid),
                    "value": views * args.steps / dt if dt > 0 else None,
            "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_ok": int(ranks_ok), "errors": [e for e in errs if e],
            "process_group": pg, "repeats": region_stats(dts, args.steps, views)[1] if dt > 0 else None,
            "mean_loss_all_ranks": loss_sum / max(ranks_ok, 1.0),
            "config": {"workload": "BASELINE configs[3] step: FORGE_poseEstimator3D GT-pose training, %d scene(s)/GPU x 5 views -> 3 fusions -> 10 rendered "
                                   "views/scene, %s, SyncBatchNorm + DDP" % (B, "reference-native 32^3 / "
                                           "64^3 grids" if args.grid == 32 else "128^3-voxel render grid from synthetic "
                                   "[128,64^3] feature volumes (encoder not run: its forward, backward and gradient all-reduce are not in "
                                           "this number)") + TRAIN_CAVEATS,
                       "perceptual_term": "excluded", "deterministic": False, "scenes_per_gpu": B, "feature_grid": args.grid,
                       "global_batch": B * world, "parallelism": "dp%d (DDP bucketed RCCL all-reduce of 221 MB fp32 gradients; HIP SyncBatchNorm)" % world,
                       "rank0_affinity": affinity}}, args.full_record)
    fdist.barrier()
    fdist.shutdown()
