"""Row f1 — loss assembly and the optimisation step around the model, mirroring the reference's
scripts/kubric_compute_loss.py (:9-42 `compute_reconstruction_loss`, :45-68 `compute_pose_loss`, :71-118 `compute_all_loss`,
:121-172 `compute_all_loss_nvs`; pinned against the reference's functions by tests/golden/loss_terms.npz),
scripts/kubric_trainer.py (:21-59: clip-norm 10 (Kubric) / 5 (OmniObject3D), gradient accumulation, optimizer step) and
utils/train_utils.py (:149-164 `adjust_lr`). Same names, argument order and return tuples, so the reference's trainer can call
them; the reference's own functions also work unchanged on the forge_amd models (they only call `model(sample, dataset, device)`).

Difference: the per-term `.item()` calls of the reference (4-8 device synchronisations per iteration) are replaced by ONE
device->host copy of the stacked loss terms.
"""
import torch
import torch.nn.functional as F


def _publish(losses, terms):
    """Logged scalars: the loss terms of all ranks averaged by ONE all-reduce of the stacked vector (RCCL over xGMI when the job is
    data-parallel; north_star "RCCL all-reduce of losses" - the reference logs rank 0's local values only), then one D2H copy.
    The gradients are not touched: DDP averages those."""
    if terms:
        from . import dist as fdist
        vec = fdist.all_reduce_mean_(torch.stack([v.detach() for v in terms.values()]))
        if vec.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture (forge_amd.graph.GraphedStep) a device->host copy is illegal: the terms stay device scalars, views of ONE
            # static tensor the replay overwrites - read them after a replay with .tolist()
            losses.update(dict(zip(terms.keys(), vec.unbind(0))))
        else:
            losses.update(dict(zip(terms.keys(), vec.cpu().tolist())))
    return losses


class _SseGroups(torch.autograd.Function):
    """Per-group sums of squared errors of rendered maps against their target views in one HIP launch (csrc/loss.hip);
    pred [B,Vp,C,H,W] any strides, target [B,Vt,C,H,W] -> [Vp / gsize] sums. Backward = one launch in pred's memory layout."""

    @staticmethod
    def forward(ctx, pred, target, gsize):
        from . import _lib
        B, Vp, C, H, W = pred.shape
        Vt = target.shape[1]
        target = target.contiguous()
        if pred.stride(0) != Vp * pred.stride(1):
            pred = pred.contiguous()
        L = _lib.lib()
        nb = L.forge_sse_groups_blocks()
        G = Vp // gsize
        partial = torch.empty(nb, G, dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(L.forge_sse_groups_fwd(_lib.ptr(pred), pred.stride(1), pred.stride(2), pred.stride(3), pred.stride(4), _lib.ptr(target),
                                              _lib.ptr(partial), B, Vp, Vt, gsize, C, H, W, _lib.current_stream()), "forge_sse_groups_fwd")
        ctx.save_for_backward(pred, target)
        ctx.gsize = gsize
        return partial.sum(dim=0)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        pred, target = ctx.saved_tensors
        B, Vp, C, H, W = pred.shape
        dpred = torch.empty_strided(pred.shape, pred.stride(), dtype=torch.float32, device=pred.device)
        coef = (2.0 * g).to(torch.float32).contiguous()
        with torch.cuda.device(pred.device):
            _lib.check(_lib.lib().forge_sse_groups_bwd(_lib.ptr(pred), pred.stride(1), pred.stride(2), pred.stride(3), pred.stride(4), _lib.ptr(target),
                                                       _lib.ptr(coef), _lib.ptr(dpred), B, Vp, target.shape[1], ctx.gsize, C, H, W,
                                                       _lib.current_stream()), "forge_sse_groups_bwd")
        return dpred, None, None


def grouped_mse(pred, target, gsize):
    """[F.mse_loss(pred[:, g*gsize:(g+1)*gsize], target[:, (g*gsize) % Vt ...]) for g] as ONE pass (f1: the four MSE terms of a training
    iteration are two calls - rgb and mask). pred [B,Vp,C,H,W], target [B,Vt,C,H,W] on the MI355X."""
    B, Vp, C, H, W = pred.shape
    if not (pred.is_cuda and pred.dtype == torch.float32):
        raise TypeError("grouped_mse: pred must be a float32 tensor on the MI355X (got %s on %s)" % (pred.dtype, pred.device))
    if target.dim() != 5 or target.shape[0] != B or tuple(target.shape[2:]) != (C, H, W):
        raise ValueError("grouped_mse: target %s does not match pred %s in batch / (C, H, W)" % (tuple(target.shape), tuple(pred.shape)))
    if Vp % int(gsize) or target.shape[1] < 1:
        raise ValueError("grouped_mse: %d predicted views are not a multiple of the group size %d" % (Vp, gsize))
    # what F.mse_loss would do implicitly: bool / uint8 masks, float64 / half images are promoted to pred's dtype, and the kernel reads
    # the target with raw fp32 pointers, so it must live on pred's device
    if target.dtype != torch.float32 or target.device != pred.device:
        target = target.to(device=pred.device, dtype=torch.float32)
    sse = _SseGroups.apply(pred, target, int(gsize))
    return sse / float(B * gsize * C * H * W)


def compute_reconstruction_loss(config, epoch, sample, dataset, model, losses, device, perceptual_loss=None):
    """GT-pose training (kubric_train_pose_3D.py): model returns 2t views per scene = [3v/2v cross views | all-view fusion]."""
    rendered_imgs, rendered_masks = model(sample, dataset, device)
    clips = sample["images"].to(device)
    masks = sample["fg_probabilities"].to(device)
    b, t, c, h, w = clips.shape
    target_imgs = clips.reshape(b * t, c, h, w)
    rendered_imgs = rendered_imgs.reshape(b, 2 * t, c, h, w)
    rendered_masks = rendered_masks.reshape(b, 2 * t, 1, h, w)
    # the four MSE terms (:26-29) as two fused passes over the rendered maps: no slicing / reshaping copies, one backward launch each
    # (grouped_mse raises on host tensors: one implementation, the CPU statement of these losses is the oracle's)
    mi, mm = grouped_mse(rendered_imgs, clips, t), grouped_mse(rendered_masks, masks, t)
    terms = {"recon_img_sv": config.loss.recon_rgb * mi[0], "recon_mask_sv": config.loss.recon_mask * mm[0],
             "recon_img_mv": config.loss.recon_rgb * mi[1], "recon_mask_mv": config.loss.recon_mask * mm[1]}
    if config.loss.perceptual_img > 0:
        tgt = target_imgs.reshape(b, t, c, h, w).repeat(1, 2, 1, 1, 1).reshape(b * 2 * t, c, h, w)
        terms["perceptual_img"] = config.loss.perceptual_img * perceptual_loss(rendered_imgs.reshape(-1, c, h, w), tgt).mean()
    loss = sum(terms.values())
    return loss, _publish(losses, terms), rendered_imgs, rendered_masks


def compute_all_loss_nvs(config, epoch, sample, dataset, model, losses, device, perceptual_loss=None):
    """Joint training (kubric_train_joint.py): 5 input + novel views, pose / translation MSE, optional origin regulariser."""
    rendered_imgs, rendered_masks, origin_proj, pose = model(sample, dataset, device)
    clips, clips_nvs = sample["images"][:, :5].to(device), sample["images"][:, 5:].to(device)
    masks, masks_nvs = sample["fg_probabilities"][:, :5].to(device), sample["fg_probabilities"][:, 5:].to(device)
    b, t, c, h, w = clips.shape
    t_all = t + clips_nvs.shape[1]
    rendered_imgs = rendered_imgs.reshape(b, t_all, c, h, w)
    rendered_masks = rendered_masks.reshape(b, t_all, 1, h, w)
    if t_all == 2 * t:
        mi = grouped_mse(rendered_imgs, sample["images"].to(device), t)          # (input views, novel views) in one pass each
        mm = grouped_mse(rendered_masks, sample["fg_probabilities"].to(device), t)
    else:                                                                         # another number of novel views: one group per call, same kernel
        n = t_all - t
        mi = (grouped_mse(rendered_imgs[:, :t], clips, t)[0], grouped_mse(rendered_imgs[:, t:], clips_nvs, n)[0])
        mm = (grouped_mse(rendered_masks[:, :t], masks, t)[0], grouped_mse(rendered_masks[:, t:], masks_nvs, n)[0])
    recon = {"recon_img": config.loss.recon_rgb * mi[0], "recon_mask": config.loss.recon_mask * mm[0],
             "recon_img_nvs": config.loss.recon_rgb * mi[1], "recon_mask_nvs": config.loss.recon_mask * mm[1]}
    terms = dict(recon, pose=F.mse_loss(pose["pred"][:, :4], pose["gt"][:, :4]), trans=F.mse_loss(pose["pred"][:, 4:], pose["gt"][:, 4:]))
    if config.loss.perceptual_img > 0:
        tgt = torch.cat([clips, clips_nvs], dim=1).reshape(b * t_all, c, h, w)
        terms["perceptual_img"] = config.loss.perceptual_img * perceptual_loss(rendered_imgs.reshape(-1, c, h, w), tgt).mean()
    if getattr(config.loss, "regu_origin_proj", 0) > 0:
        terms["regu_origin"] = config.loss.regu_origin_proj * F.mse_loss(origin_proj, torch.full_like(origin_proj, 0.5))
    loss = sum(terms.values())
    return loss, _publish(losses, terms), rendered_imgs, rendered_masks


def compute_pose_loss(config, epoch, sample, dataset, model, losses, device, perceptual_loss=None):
    """Pose-only training (scripts/kubric_compute_loss.py:45-68: `parameter` in {'pose', 'pose_head'}): model returns
    (pose dict, origin projection). The reference's origin regulariser branch (epoch >= 100) reads an undefined variable; here it is the
    same term `compute_all_loss` uses (target (0.5, 0.5))."""
    pose, origin_proj = model(sample, dataset, device)
    terms = {
        "pose": F.mse_loss(pose["pred"][:, :4], pose["gt"][:, :4]),
        "trans": F.mse_loss(pose["pred"][:, 4:], pose["gt"][:, 4:]),
    }
    if getattr(config.loss, "regu_origin_proj", 0) > 0 and epoch >= 100:
        terms["regu_origin"] = config.loss.regu_origin_proj * F.mse_loss(origin_proj, torch.full_like(origin_proj, 0.5))
    loss = sum(terms.values())
    return loss, _publish(losses, terms), None, None


def compute_all_loss(config, epoch, sample, dataset, model, losses, device, perceptual_loss=None):
    """Reconstruction + pose loss with the 2t-view layout of the GT-pose model (scripts/kubric_compute_loss.py:71-118)."""
    rendered_imgs, rendered_masks, origin_proj, pose = model(sample, dataset, device)
    clips = sample["images"].to(device)
    masks = sample["fg_probabilities"].to(device)
    b, t, c, h, w = clips.shape
    target_imgs = clips.reshape(b * t, c, h, w)
    rendered_imgs = rendered_imgs.reshape(b, 2 * t, c, h, w)
    rendered_masks = rendered_masks.reshape(b, 2 * t, 1, h, w)
    mi, mm = grouped_mse(rendered_imgs, clips, t), grouped_mse(rendered_masks, masks, t)
    terms = {
        "recon_img_sv": config.loss.recon_rgb * mi[0], "recon_mask_sv": config.loss.recon_mask * mm[0],
        "recon_img_mv": config.loss.recon_rgb * mi[1], "recon_mask_mv": config.loss.recon_mask * mm[1],
        "pose": F.mse_loss(pose["pred"][:, :4], pose["gt"][:, :4]),
        "trans": F.mse_loss(pose["pred"][:, 4:], pose["gt"][:, 4:]),
    }
    if config.loss.perceptual_img > 0:
        tgt = target_imgs.reshape(b, t, c, h, w).repeat(1, 2, 1, 1, 1).reshape(b * 2 * t, c, h, w)
        terms["perceptual_img"] = config.loss.perceptual_img * perceptual_loss(rendered_imgs.reshape(-1, c, h, w), tgt).mean()
    if getattr(config.loss, "regu_origin_proj", 0) > 0:
        terms["regu_origin"] = config.loss.regu_origin_proj * F.mse_loss(origin_proj, torch.full_like(origin_proj, 0.5))
    loss = sum(terms.values())
    return loss, _publish(losses, terms), rendered_imgs, rendered_masks


def adjust_lr(config, optimizer, iter_num, adjust_iter_num):
    """utils/train_utils.py:149-164: lr = base * 0.5^k at the k-th milestone (OmniObject3D: linear warm-up over 500 iterations)."""
    lr = None
    if config.dataset.name == "omniobject3d":
        lr = config.train.lr * iter_num / 500
    for k, it in enumerate(adjust_iter_num[:4]):
        if iter_num == it:
            lr = config.train.lr * 0.5 ** (k + 1)
    if lr is not None:
        for group in optimizer.param_groups:
            group["lr"] = lr
    return lr


def enable_ray_sharding(model, on=True, group=None, reduce="all"):
    """BASELINE configs[4] ("8 GPUs with per-ray sharding"): every rank processes the SAME batch, the ray-march of every rendered view
    is split into row bands over the ranks of `group` (forge_amd/dist.py::render_rays_sharded: all_gather of the maps forward, all-reduce of
    d(volume) / d(cameras) backward), encoder / pose networks / fusion / conv_rgb run replicated. Works on the bare model or a DDP wrapper
    (DDP then averages identical gradients, which also keeps the replicas bit-identical despite atomics-ordered weight gradients)."""
    m = model.module if hasattr(model, "module") else model
    m.render.ray_shard, m.render.ray_shard_group, m.render.ray_shard_reduce = bool(on), group, reduce
    return model


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm, norm_type=2) (scripts/kubric_trainer.py:56) on the multi-tensor (foreach) kernels.
    torch's own foreach path does not get there for FORGE: the 2-D pose estimator's positional embedding is a float64 parameter
    (models/pose_estimator_2d.py:50-51), so the total norm - and with it the clip coefficient - is a float64 tensor, `_foreach_mul_` refuses a scalar
    tensor of another dtype than the gradients and falls back to one type-promoting `mul_` per gradient: 550 launches, 2.6 ms per joint step
    (tools/debug/joint_clip_probe.py). Same arithmetic here - per-tensor norms in the gradients' dtype, their 2-norm in the promoted dtype
    (as torch.nn.utils.get_total_norm), coefficient clamped to 1 - with the coefficient cast to each dtype group before ONE multi-tensor multiply per
    group, which is the conversion the per-tensor `mul_` performs on the scalar operand anyway. Returns the total norm."""
    params = [p for p in ([parameters] if torch.is_tensor(parameters) else parameters) if p.grad is not None]
    grads = [p.grad for p in params]
    if not grads:
        return torch.zeros(())
    with torch.no_grad():
        groups = {}
        for g in grads:
            groups.setdefault((g.device, g.dtype), []).append(g)
        # torch.nn.utils.get_total_norm: per-tensor norms group by group (multi-tensor kernels), stacked in that order in the promoted dtype, then
        # the 2-norm of the stack. The stack itself is written by ONE multi-tensor copy per dtype group into a preallocated vector (then cast and joined): torch.stack of 551 zero-dim
        # tensors is a 1.5 ms chain of batched-cat launches (tools/joint_op_profile.py). Same values in the same order -> the same vector_norm.
        first, dtype, parts = grads[0].device, grads[0].dtype, []
        for (device, gdtype), gs in groups.items():
            vec = torch.empty(len(gs), dtype=gdtype, device=device)
            torch._foreach_copy_(list(vec.unbind(0)), torch._foreach_norm(gs, 2.0))       # same dtype on both sides: the multi-tensor route
            parts.append(vec.to(first))
            dtype = torch.promote_types(dtype, gdtype)
        stacked = parts[0].to(dtype) if len(parts) == 1 else torch.cat([v.to(dtype) for v in parts])
        total = torch.linalg.vector_norm(stacked, 2.0)
        coef = torch.clamp(float(max_norm) / (total + 1e-6), max=1.0)
        for (device, gdtype), gs in groups.items():
            torch._foreach_mul_(gs, coef.to(device=device, dtype=gdtype))
    return total


def train_step(config, sample, dataset, model, optimizer, device, loss_func=compute_reconstruction_loss, epoch=0, batch_idx=0,
               perceptual_loss=None):
    """One iteration of scripts/kubric_trainer.py:47-59 (without the logging): loss, backward, clip, optimizer step. With ray sharding
    enabled (enable_ray_sharding) rank 0's sample is broadcast first, so that all ranks render the same batch."""
    max_norm = 5.0 if config.dataset.name == "omniobject3d" else 10.0
    bare = model.module if hasattr(model, "module") else model
    if getattr(getattr(bare, "render", None), "ray_shard", False):
        from . import dist as fdist
        # the group's first rank owns the batch (a GLOBAL rank, as torch.distributed.broadcast wants it: sub-groups need not contain rank 0)
        sample = fdist.broadcast_sample({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()}, src=None, group=bare.render.ray_shard_group)
    accumulation = getattr(config.train, "accumulation_step", 1)
    loss, losses, imgs, masks = loss_func(config, epoch, sample, dataset, model, {}, device, perceptual_loss)
    (loss / accumulation).backward()
    if (batch_idx + 1) % accumulation == 0:
        clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
        optimizer.zero_grad()
    return loss.detach(), losses
