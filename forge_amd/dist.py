"""Single-node data parallelism: one process per MI355X, torch.distributed over RCCL (backend
"nccl" IS RCCL on PyTorch-ROCm) - the only parallelism the reference has (SURVEY.md 2.2:
DDP + DistributedSampler, kubric_train_pose_3D.py:74,124,130; eval sharding by batch index,
kubric_eval.py:56). Scenes are independent, so the default data path needs NO collective: scenes are
sharded across ranks and only scalar metrics (losses / SSE / counts / time) are all-reduced;
gradient all-reduce in training is torch DDP's bucketed RCCL all-reduce, unchanged.
Second half of the file: per-RAY sharding of one batch (BASELINE configs[4]) - differentiable row-band
ray-march with an all_gather forward and an all-reduce / reduce-to-owner of d(volume) backward - and the
cross-rank BatchNorm statistics exchange used by forge_amd.fusion's HIP SyncBatchNorm.
"""
import os
import sys

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None, allow_shared_gpus=False, timeout_s=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1).
    Returns (rank, local_rank, world_size). With fewer visible GPUs than ranks RCCL cannot run (it refuses two ranks on one device): that
    is an ERROR unless the caller asks for a functional rehearsal with `allow_shared_gpus=True` (ranks share devices, collectives on gloo) -
    a scaling number from shared devices would be meaningless, so nothing drops to gloo silently. An explicit `backend` is taken as is."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl" and torch.cuda.device_count() < world:
                if not allow_shared_gpus:
                    raise RuntimeError("forge_amd.dist.init: %d ranks but only %d GPU(s) visible - one process per MI355X is the deployment; pass "
                                       "allow_shared_gpus=True for a functional rehearsal on shared devices (collectives on gloo)" % (world, torch.cuda.device_count()))
                print("forge_amd.dist: %d ranks on %d GPU(s) - rehearsal on shared devices, collectives on gloo"
                      % (world, torch.cuda.device_count()), file=sys.stderr)
                backend = "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        kw = {}
        if timeout_s:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def group_info():
    """What carried (or would carry) this process's collectives: backend of the default process group ("nccl" = RCCL over xGMI on ROCm,
    "gloo" for rehearsals with more ranks than GPUs), its size and this rank's device - printed in every bench line so that a multi-GPU
    record shows which library saw how many ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"initialized": False, "backend": None, "world_size": 1}
    info = {"initialized": True, "backend": str(dist.get_backend()), "world_size": dist.get_world_size(), "rank": dist.get_rank()}
    if torch.cuda.is_available():
        info["device"] = "cuda:%d" % torch.cuda.current_device()
        info["visible_gpus"] = torch.cuda.device_count()
    return info


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


def gather_strings(msg):
    """Every rank's (short) message or None on every rank - per-rank failure reports of the bench harness. Uses all_gather_object (pickled
    through the backend); the list is [msg] for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, msg)
        return out
    return [msg]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl" and torch.cuda.is_available():
            dist.barrier(device_ids=[torch.cuda.current_device()])      # RCCL: barrier on THIS rank's GPU (not a guess from the rank number)
        else:
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


def _group_info(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _gather_rows(band, world, group):
    """[V, C, band, Wr] per rank -> [V, C, world * band, Wr] on every rank (ONE all_gather; rank r's rows land at [r band, (r+1) band))."""
    band = band.contiguous()
    parts = [torch.empty_like(band) for _ in range(world)]
    dist.all_gather(parts, band, group=group)
    return torch.cat(parts, dim=2)


def _dense_view(g):
    """A contiguous view of g's memory for the collective (channels-last volumes are dense but not `is_contiguous()`); a copy otherwise."""
    if g.is_contiguous():
        return g
    if g.dim() == 5 and g.permute(0, 2, 3, 4, 1).is_contiguous():
        return g.permute(0, 2, 3, 4, 1)
    return g.contiguous()


class _RenderRaysSharded(torch.autograd.Function):
    """Differentiable ray-sharded render (BASELINE configs[4]).
      forward   every rank marches its band of image rows of every view (band_cameras: principal point shifted by the band's first
                row, so the unchanged ray-march kernel produces exactly those rows); ONE all_gather per output assembles the images.
      backward  the rank takes the rows of the incoming image gradients that belong to ITS band, runs the ray-march backward
                (forge_render_bwd) on them -> partial d(features), d(density) [the size of the volume: 17.8 MB at 64^3, 142.6 MB at
                128^3] and d(cameras); `reduce="all"`: the partials are summed over ranks (all_reduce - every rank continues the
                backward through its replica of fusion / encoder with the full gradient); `reduce="none"`: the partials are returned
                as they are (the caller's differentiable broadcast_from_owner reduces them to the volume's owner instead).
    The band render runs under a private autograd graph (enable_grad on detached leaves), so any `render_fn` with ops.render_rays'
    signature works - the HIP op in the product, the CPU oracle in the gloo tests."""

    @staticmethod
    def forward(ctx, feat, dens, cam, view2vol, render_fn, cfg, group, reduce):
        Hr, Wr, S, zmin, zmax, half, want_depth = cfg
        rank, world = _group_info(group)
        h0, h1 = ray_band(Hr, rank, world)
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]]
        with torch.enable_grad():
            f = feat.detach().requires_grad_(need[0])
            d = dens.detach().requires_grad_(need[1])
            c = band_cameras(cam.detach(), h0).requires_grad_(need[2])      # cy' = cy - h0: d/d cam = d/d band camera
            outs = render_fn(f, d, c, view2vol, h1 - h0, Wr, S, zmin, zmax, half, want_depth)
        ctx.local = (f, d, c, outs) if any(need) else None
        ctx.meta = (h0, h1, world, group, reduce, need)
        full = tuple(_gather_rows(o.detach(), world, group) if world > 1 else o.detach() for o in outs)
        return full

    @staticmethod
    def backward(ctx, *grads):
        h0, h1, world, group, reduce, need = ctx.meta
        f, d, c, outs = ctx.local
        gb = [torch.zeros_like(o) if g is None else g[:, :, h0:h1].contiguous() for g, o in zip(grads, outs)]
        leaves = [t for t, n in zip((f, d, c), need) if n]
        got = iter(torch.autograd.grad(outs, leaves, gb, allow_unused=True))
        res = []
        for t, n in zip((f, d, c), need):
            g = next(got) if n else None
            res.append(torch.zeros_like(t) if (n and g is None) else g)
        ctx.local = None
        if world > 1 and reduce == "all":
            views = [None if g is None else _dense_view(g) for g in res]
            work = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in views if v is not None]
            for w in work:
                w.wait()
            # _dense_view had to copy (a gradient that is neither contiguous nor channels-last, e.g. an expanded one): the copy - same shape - IS the sum
            res = [g if (g is None or v.data_ptr() == g.data_ptr()) else v for g, v in zip(res, views)]
        return res[0], res[1], res[2], None, None, None, None, None


def render_rays_sharded(feat, dens, cam, view2vol, Hr, Wr, S, zmin, zmax, half, want_depth=False, render_fn=None, group=None, reduce="all"):
    """Ray-sharded ops.render_rays, differentiable. Every rank passes the SAME arguments (replicated volume and cameras) and receives the
    full-size outputs; gradients w.r.t. feat / dens / cam are the single-process ones on every rank (reduce="all") or this rank's band
    partials (reduce="none", for broadcast_from_owner). `render_fn` defaults to the HIP op (tests inject the CPU oracle)."""
    if render_fn is None:
        from . import ops
        render_fn = ops.render_rays
    if reduce not in ("all", "none"):
        raise ValueError("reduce must be 'all' or 'none'")
    cfg = (int(Hr), int(Wr), int(S), float(zmin), float(zmax), tuple(float(h) for h in half), bool(want_depth))
    return _RenderRaysSharded.apply(feat, dens, cam, view2vol, render_fn, cfg, group, reduce)


class _BroadcastFromOwner(torch.autograd.Function):
    """Differentiable broadcast of the owner's tensors ("encoder + fusion for a scene run on its owner rank, fused volume
    broadcast once ... backward: each rank's d(volume) partials reduce to the owner"). Forward: rank `src`'s tensors reach every rank;
    backward: the ranks' gradients are summed onto `src` (dist.reduce), the other ranks get zeros (their inputs were placeholders)."""

    @staticmethod
    def forward(ctx, src, group, *tensors):
        ctx.src, ctx.group = src, group
        rank, world = _group_info(group)
        outs = []
        for t in tensors:
            o = t.detach().clone(memory_format=torch.preserve_format)          # channels-last volumes stay channels-last (no layout round trip)
            if world > 1:
                v = _dense_view(o)
                dist.broadcast(v, src=src, group=group)
                if v.data_ptr() != o.data_ptr():
                    o.copy_(v)
            outs.append(o)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        rank, world = _group_info(ctx.group)
        res = []
        for g in grads:
            if g is None:
                res.append(None)
                continue
            g = g.clone(memory_format=torch.preserve_format)
            if world > 1:
                v = _dense_view(g)
                dist.reduce(v, dst=ctx.src, op=dist.ReduceOp.SUM, group=ctx.group)
                if v.data_ptr() != g.data_ptr():
                    g.copy_(v)
                if rank != ctx.src:
                    g.zero_()
            res.append(g)
        return (None, None) + tuple(res)


def broadcast_from_owner(tensors, src, group=None):
    """tensors (tuple) of the owner rank `src` (GLOBAL rank, as torch.distributed.broadcast takes it) -> the same values on every rank,
    differentiable (gradients are reduced to the owner). Non-owners pass placeholders of the same shape / dtype / device / memory format."""
    return _BroadcastFromOwner.apply(int(src), group, *tensors)


def broadcast_sample(sample, src=None, group=None):
    """Ray-sharded training renders ONE batch on all ranks: rank `src`'s sample dict (tensors of equal shapes on every rank) is broadcast.
    src is a GLOBAL rank (torch.distributed.broadcast's convention); None = the first rank of `group` (rank 0 of the default group)."""
    rank, world = _group_info(group)
    if world == 1:
        return sample
    if src is None:
        src = dist.get_global_rank(group, 0) if group is not None else 0
    out = {}
    for k in sorted(sample):
        v = sample[k]
        if torch.is_tensor(v):
            v = v.contiguous().clone()
            dist.broadcast(v, src=src, group=group)
        out[k] = v
    return out
