"""Single-node data parallelism: one process per MI355X, torch.distributed over RCCL (backend
"nccl" IS RCCL on PyTorch-ROCm) — the only parallelism the reference has (SURVEY.md §2.2:
DDP + DistributedSampler, kubric_train_pose_3D.py:74,124,130; eval sharding by batch index,
kubric_eval.py:56). Scenes are independent, so the data path needs NO collective: scenes are
sharded across ranks and only scalar metrics (losses / SSE / counts / time) are all-reduced.
Gradient all-reduce in training is torch DDP's bucketed RCCL all-reduce, unchanged.
"""
import os
import sys

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1).
    Returns (rank, local_rank, world_size)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl" and torch.cuda.device_count() < world:
                # more ranks than GPUs (a multi-rank dry run on a 1-GPU box): RCCL refuses two ranks on one device, gloo does not
                print("forge_amd.dist: %d ranks on %d GPU(s) - falling back to gloo, ranks share devices"
                      % (world, torch.cuda.device_count()), file=sys.stderr)
                backend = "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n_items, rank, world):
    """Round-robin shard of scene indices (kubric_eval.py:56: `batch_idx % split_num == exp_id`)."""
    return list(range(rank, n_items, world))


def all_reduce_scalars(values, device, op="sum"):
    """All-reduce a small list of python floats (SSE, counts, seconds...). float64 on the wire."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
    return t.tolist()


def all_reduce_mean_(t):
    """In-place mean over ranks of a small device tensor (loss terms / metrics); identity for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= dist.get_world_size()
    return t


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    """Tear the default process group down (clean exit of the RCCL communicator); no-op for world size 1."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def psnr_from_sse(sse, count):
    """10 log10(1 / MSE), data_range = 1 (utils/eval_utils.py:8-12)."""
    import math
    mse = sse / max(count, 1.0)
    return float("inf") if mse <= 0 else 10.0 * math.log10(1.0 / mse)


# ---------------------------------------------------------------------------------------------------------------------
# Per-ray sharding of ONE scene's views across ranks (BASELINE configs[4]: "8 GPUs with per-ray sharding").
# The fused volume is replicated (broadcast by the caller, 17.8 MB at 64^3); every rank marches a contiguous band of image
# rows of every view — rendering rows [h0, h1) equals rendering a full image whose principal point is shifted by h0 — and the
# bands are all-gathered (RCCL). Worth it only when one scene must be rendered faster than one GPU can (the ray-march is ~1 %
# of the reconstruction step); scene sharding above is the default.
# ---------------------------------------------------------------------------------------------------------------------
def ray_band(Hr, rank, world):
    """Contiguous row band [h0, h1) of rank `rank`; requires Hr % world == 0 (equal bands -> one all_gather)."""
    if Hr % world:
        raise ValueError("render height %d is not divisible by world size %d" % (Hr, world))
    band = Hr // world
    return rank * band, (rank + 1) * band


def band_cameras(cam, h0):
    """cam [V,16] (R9, T3, fx, fy, cx, cy): the same cameras seen through the window starting at row h0 (cy -> cy - h0)."""
    out = cam.clone()
    out[:, 15] = out[:, 15] - float(h0)
    return out


def render_rays_sharded(feat, dens, cam, view2vol, Hr, Wr, S, zmin, zmax, half, want_depth=False, render_fn=None, group=None):
    """Ray-sharded version of ops.render_rays (inference). Every rank passes the SAME arguments; returns the full-size outputs on
    every rank. `render_fn` defaults to the HIP op (tests inject the CPU oracle)."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if render_fn is None:
        from . import ops
        render_fn = ops.render_rays
    h0, h1 = ray_band(Hr, rank, world)
    outs = render_fn(feat, dens, band_cameras(cam, h0), view2vol, h1 - h0, Wr, S, zmin, zmax, half, want_depth)
    if world == 1:
        return outs
    full = []
    for o in outs:                                             # [V, C, band, Wr] -> gather along the row axis
        o = o.contiguous()
        parts = [torch.empty_like(o) for _ in range(world)]
        dist.all_gather(parts, o, group=group)
        full.append(torch.cat(parts, dim=2))
    return tuple(full)
