"""Sub-records of the `--gpus N` line whose collectives matter: DDP + SyncBatchNorm training (configs[3]), ray-sharded joint step (configs[4])."""
import os

import torch

from forge_amd import dist as fdist, synthetic as syn
from benchkit.common import T_IN, TRAIN_CAVEATS, _bracketed
from benchkit.emit import emit


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
        sample["features_recon"] = torch.randn(scenes, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5,
                2).contiguous().permute(0, 1, 5, 2, 3, 4)
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
                                                                    "128^3-voxel render grid from synthetic [128,64^3] feature volumes "
                                                                            "(encoder not run)") + TRAIN_CAVEATS,
            "perceptual_term": "excluded", "deterministic": False, "scenes_per_gpu": scenes, "global_batch": scenes * world, "feature_grid": grid,
                    "steps": steps,
            "ms_per_step": s_sync * 1e3, "views_per_s": scenes * 10 * world / s_sync, "ms_per_step_no_sync": s_nosync * 1e3,
            "gradient_all_reduce_exposed_ms": (s_sync - s_nosync) * 1e3,
            "gradient_bytes_all_reduced_per_step": grad_bytes, "syncbn_layers": n_bn,
            "syncbn_bytes_all_reduced_per_step": (2 * bn_ch + n_bn) * 8 + 2 * bn_ch * 8,
            "mean_loss_all_ranks": fdist.all_reduce_scalars([l_sync], dev, "sum")[0] / world,
            "note": "no_sync = the same step without DDP's gradient all-reduce (SyncBatchNorm statistics still exchanged): the difference is the "
                    "all-reduce time "
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
    # kubric_train_joint.py:136 (HIP SyncBatchNorm: one all-reduce per layer and direction)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model.to(dev).train())
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True)         # kubric_train_joint.py:141
    params = [p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head,
            model.render) for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    # the same scene on every rank (train_step broadcasts rank 0's anyway)
    sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}
    if grid == 64:
        gen = torch.Generator(device=dev).manual_seed(79)
        sample["features_recon"] = torch.randn(1, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5,
                2).contiguous().permute(0, 1, 5, 2, 3, 4)
    ds = syn.SyntheticDataset(1.5)
    loss = [None]

    def step():
        loss[0] = train.train_step(cfg, sample, ds, ddp, opt, dev, loss_func=train.compute_all_loss_nvs)[0]
    out = {"workload": "BASELINE configs[4] step: FORGE joint 2D3D fine-tune (predicted poses), 1 scene x 5 input + 5 novel views -> 10 rendered views, "
            "rays of every "
                       "view sharded over the ranks in row bands; %s; DDP over "
                               "the replicas" % ("reference-native grids" if grid == 32 else "128^3-voxel render grid (synthetic "
                                       "64^3 features)") + TRAIN_CAVEATS,
           "perceptual_term": "excluded", "deterministic": False,
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
        cam = torch.cat([extr[:, :3, :3].reshape(10, 9), extr[:, :3, 3], K[0, 0].expand(10, 1), K[1, 1].expand(10, 1), K[0, 2].expand(10, 1), K[1,
                2].expand(10, 1)],
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
            emit(dict(minimal, multi_rank={"error": "sub-records did not finish within %.0f s; main line printed by the watchdog" % deadline}),
                 getattr(args, "full_record", None))
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
