"""Row f2 — test-time pose optimisation (kubric_eval.py:412-530 `do_refinement`, demo.py:115-188 `refine_pose`): thousands of
forward+backward passes through rotate -> fuse -> heads -> render w.r.t. the 7-D relative poses (quaternion + translation) of the
non-reference views, features frozen. This is the reference's largest wall-clock consumer (5000 iterations per instance by default).

Every kernel of the loop is a hand-written HIP kernel: the pose gradient flows through forge_render_bwd (camera gradients),
forge_rotate_bwd (affine gradients) and the data-gradient GEMMs of the ConvGRU / heads; weight gradients are switched off.
Same optimiser as the reference (Adam, lr 1e-3 rotation / 5e-4 translation, ExponentialLR with gamma = 1), same loss
(recon_rgb * MSE(rgb) + recon_mask * MSE(mask)); the per-iteration host-side pose metric (`.cpu()` every iteration in the
reference, kubric_eval.py:508-517) is evaluated only every `log_every` iterations. The fixed-shape iteration is captured into a
hipGraph and replayed (`use_graph`).
"""
import time

import torch
import torch.nn.functional as F

from . import geo_utils
from .model import chose_selected, sequence_from_distance


def _render_views(model, config, dataset, features, pose7, K, device, canonical=None):
    """The piecewise evaluation calls of kubric_eval.py:456-491 for given 7-D poses (quat2mat, canonical @ rel, rotate, reorder, fuse, heads,
    render with depth): what callers use to render targets / evaluate refined poses. The optimisation loop itself runs _render_views_fused."""
    b, t = features.shape[:2]
    D = features.shape[3]
    rel = model.encoder_traj.toSE3(pose7)                                              # [b(t-1),4,4]
    if canonical is None:                                                              # (pose, extrinsics) of the reference view, on `device`
        canonical = (dataset.get_canonical_pose_cv2(device=device), dataset.get_canonical_extrinsics_cv2(device=device))
    can_p, can_e = canonical
    poses = (can_p.unsqueeze(0) @ rel)
    extr = geo_utils.inverse_affine(poses).reshape(b, t - 1, 4, 4)                      # closed form: no host sync (torch.inverse has one)
    poses = torch.cat([can_p.reshape(1, 1, 4, 4).repeat(b, 1, 1, 1), poses.reshape(b, t - 1, 4, 4)], dim=1)
    extr = torch.cat([can_e.reshape(1, 1, 4, 4).repeat(b, 1, 1, 1), extr], dim=1)
    ft = model.rotate(voxels=features, camPoses_cv2=poses, grid_size=D)
    ft = chose_selected(ft, sequence_from_distance(poses[:, :, :3, 3]))
    fused = model.encoder_3d.fuse(ft)
    feat, dens = model.encoder_3d.heads(fused)
    cams = {"R": extr.reshape(b * t, 4, 4)[:, :3, :3], "T": extr.reshape(b * t, 4, 4)[:, :3, 3], "K": K.reshape(b * t, 3, 3)}
    v2v = torch.arange(b, device=device, dtype=torch.int32).repeat_interleave(t)
    imgs, masks, depths, origin = model.render(cams, feat, dens, return_origin_proj=True, render_depth=True, view2vol=v2v)
    return imgs, masks, depths, origin, poses


class _SmallAdam:
    """torch.optim.Adam's update (defaults: betas (0.9, 0.999), eps 1e-8, no weight decay) for a few tiny tensors: ONE launch per tensor
    (forge_adam_small, step count on the device) instead of the ~30 of the capturable torch optimiser. groups = [(tensor, lr), ...]."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8):
        self.groups, self.betas, self.eps = [(p, float(lr)) for p, lr in groups], betas, eps
        self.state = [(torch.zeros_like(p), torch.zeros_like(p), torch.zeros(1, dtype=torch.float32, device=p.device)) for p, _ in self.groups]

    def zero_grad(self, set_to_none=True):
        for p, _ in self.groups:
            p.grad = None

    def step(self):
        from . import _lib
        for (p, lr), (m, v, k) in zip(self.groups, self.state):
            if p.grad is None:
                continue
            with torch.cuda.device(p.device):
                _lib.check(_lib.lib().forge_adam_small(_lib.ptr(p.data), _lib.ptr(p.grad.contiguous()), _lib.ptr(m), _lib.ptr(v), _lib.ptr(k), p.numel(), lr,
                                                       self.betas[0], self.betas[1], self.eps, _lib.current_stream()), "forge_adam_small")


def _render_views_fused(model, config, dataset, features, rot, trans, K, device, canonical, const0=None):
    """_render_views with the pose algebra on ONE HIP launch (ops.pose_chain: raw quaternion / translation -> warp affine + packed cameras, Jacobian
    by forward-mode differentiation) instead of ~250 torch launches per iteration; what the refinement loop runs. Same outputs."""
    from . import ops
    b, t = features.shape[:2]
    C, D = features.shape[2], features.shape[3]
    can_p, can_e = canonical
    xf, cam, mode, slot, poses, origin = ops.pose_chain(rot, trans, can_p, can_e, K, model.rotate.half_extent(D), b, t)
    ft = ops.rotate_warp(features.reshape(b * t, C, D, D, D), xf, mode, slot).reshape(b, t, C, D, D, D)       # stored in sequence_from_distance's order
    # slot 0 always holds view 0 (distance 0, ties broken by index): the fixed reference view, copied un-warped from frozen features - nothing
    # differentiable lies behind it, so the fusion's backward skips the input-half data gradients of that step
    # (and, with frozen features, it holds the same values in every iteration: const0, the caller's dict, keeps its input-half products)
    fused = model.encoder_3d.fuse(ft, skip_dx0=not features.requires_grad, const0=None if features.requires_grad else const0)
    feat, dens = model.encoder_3d.heads(fused)
    v2v = torch.arange(b, device=device, dtype=torch.int32).repeat_interleave(t)
    imgs, masks, depths, origin = model.render({"packed": cam, "origin": origin}, feat, dens, return_origin_proj=True, render_depth=True, view2vol=v2v)
    return imgs, masks, depths, origin, poses


class PoseRefiner:
    """One instance of the refinement problem (kubric_eval.py:412-530): features [b,t,C,D,H,W] (detached encoder output), initial poses
    [b(t-1),7] (quat, trans), targets, intrinsics. `iteration()` = forward, loss, backward through every HIP kernel, Adam step; `capture()`
    records it into a hipGraph on `stream` (fixed shapes, no host synchronisation: closed-form pose inverses, device-side view ordering,
    capturable Adam), `step()` replays it (or runs it eagerly). The model's weights must be frozen by the caller (refine_poses does)."""

    def __init__(self, model, config, dataset, features, poses_cam, target_imgs, target_masks, K, device, use_graph=True, stream=None):
        self.model, self.config, self.dataset, self.device = model, config, dataset, device
        self.features = features.detach().to(device)
        self.target_imgs, self.target_masks, self.K = target_imgs.to(device), target_masks.to(device), K.to(device)
        self.canonical = (dataset.get_canonical_pose_cv2(device=device), dataset.get_canonical_extrinsics_cv2(device=device))
        self.rot = poses_cam[:, :4].detach().clone().to(device).requires_grad_(True)
        self.trans = poses_cam[:, 4:].detach().clone().to(device).requires_grad_(True)
        if not self.features.is_cuda:
            raise RuntimeError("forge_amd: PoseRefiner runs only on the MI355X HIP kernels (features on %s); there is no CPU path" % self.features.device)
        lr = 0.001
        self.opt = _SmallAdam([(self.rot, lr), (self.trans, lr / 2.0)])
        self.w_rgb, self.w_mask = config.loss.recon_rgb, config.loss.recon_mask     # (the reference's ExponentialLR has gamma = 1: a constant rate)
        self.use_graph, self.graph, self.static_loss = bool(use_graph), None, None
        self.stream = stream
        # iteration-invariant products of the reference view (slot 0: frozen features, fixed pose), filled by the first forward, valid for THIS
        # instance's features and the model's frozen weights (ConvGRU_3D.fuse_frozen_hip)
        self.fuse_const = {}

    def iteration(self):
        imgs, masks, _, _, _ = _render_views_fused(self.model, self.config, self.dataset, self.features, self.rot, self.trans, self.K, self.device,
                                                   self.canonical, const0=self.fuse_const)
        loss = self.w_rgb * F.mse_loss(imgs, self.target_imgs) + self.w_mask * F.mse_loss(masks, self.target_masks)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def eager_step(self):
        self.opt.zero_grad(set_to_none=True)
        return self.iteration()

    def capture(self):
        if not self.fuse_const:                          # the hoisted products must exist BEFORE the capture (else they would be recomputed,
            with torch.no_grad():                        # inside the graph's pool, by every replay): one forward pass, no optimiser step
                _render_views_fused(self.model, self.config, self.dataset, self.features, self.rot, self.trans, self.K, self.device, self.canonical,
                                    const0=self.fuse_const)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)             # gradients are (re)allocated inside the graph's private pool
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_loss = self.iteration()          # the capture pass does not execute: the first replay is the first iteration

    def step(self):
        """One optimisation iteration; on self.stream when one was given (several refiners in flight: refine_poses_many)."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                return self._step()
        return self._step()

    def _step(self):
        if self.graph is not None:
            self.graph.replay()
            return self.static_loss
        return self.eager_step()

    def poses(self):
        return torch.cat([F.normalize(self.rot), self.trans], dim=1).detach()


def _frozen(model):
    frozen = [p for p in model.parameters() if p.requires_grad]
    for p in frozen:                                   # no weight gradients: only the poses are optimised
        p.requires_grad_(False)
    return frozen


def refine_poses(model, config, dataset, features, poses_cam, target_imgs, target_masks, K, device, iter_num=500, log_every=0,
                 use_graph=True):
    """features [b,t,C,D,H,W] (detached encoder output), poses_cam [b(t-1),7] initial (quat, trans), target_imgs [b*t,3,H,W],
    target_masks [b*t,1,H,W], K [b,t,3,3]. Returns (refined poses [b(t-1),7], list of losses, seconds per iteration).

    use_graph: after a few eager iterations (weight packing, allocator warm-up) ONE iteration - forward, loss, backward through every
    HIP kernel, Adam step - is captured into a hipGraph and replayed; the loop body is fixed-shape and free of host synchronisation
    (closed-form pose inverses, device-side view ordering, capturable Adam), so the replay does exactly what the eager iteration does."""
    model.eval()
    frozen = _frozen(model)
    try:
        r = PoseRefiner(model, config, dataset, features, poses_cam, target_imgs, target_masks, K, device, use_graph=use_graph)
        history = []
        warm = min(3, iter_num)
        t0 = None
        for it in range(iter_num + 1):
            if it == warm:                              # time the steady state (first iterations load kernels / warm the allocator)
                if use_graph:
                    r.capture()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
            loss = r.step() if it >= warm else r.eager_step()
            if log_every and it % log_every == 0:
                history.append(loss.item())
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / max(iter_num + 1 - warm, 1)
        return r.poses(), history, dt
    finally:
        for p in frozen:
            p.requires_grad_(True)


def refine_poses_many(model, config, dataset, problems, device, iter_num=500, depth=2):
    """Several refinement problems (kubric_eval.py refines every test instance independently) with `depth` of them IN FLIGHT: each problem's
    iteration is its own hipGraph on its own HIP stream and the replays are issued round-robin, so that one instance's latency-bound
    launches (pose algebra, rotate, heads, ray-march backward) share the chip with another's GEMMs - the refinement counterpart of
    graph.PipelinedForward. problems = list of (features, poses_cam, target_imgs, target_masks, K). Returns ([refined poses], seconds per
    iteration AND instance)."""
    model.eval()
    frozen = _frozen(model)
    try:
        done, dt_total, n_iter = [], 0.0, 0
        for g0 in range(0, len(problems), depth):
            group = problems[g0:g0 + depth]
            cur = torch.cuda.current_stream(device)
            refs = [PoseRefiner(model, config, dataset, *pr, device, use_graph=True, stream=torch.cuda.Stream(device=device)) for pr in group]
            warm = min(3, iter_num)
            for r in refs:                               # warm-up + capture on the caller's stream, one refiner at a time
                for _ in range(warm):
                    r.eager_step()
                r.capture()
            torch.cuda.synchronize(device)
            for r in refs:
                r.stream.wait_stream(cur)
            t0 = time.perf_counter()
            replays = iter_num + 1 - warm                 # iter_num + 1 optimiser steps in total, as refine_poses and kubric_eval.py:450 (range(iter_num + 1))
            for _ in range(replays):
                for r in refs:
                    r.step()
            torch.cuda.synchronize(device)
            dt_total += time.perf_counter() - t0
            n_iter += replays * len(refs)
            done.extend(r.poses() for r in refs)
        return done, dt_total / max(n_iter, 1)
    finally:
        for p in frozen:
            p.requires_grad_(True)


def pose_errors(pred7, gt44):
    """rotation error (deg, quaternion angle as utils/eval_utils.py:14-33) and translation error (L2) per pose"""
    gq = geo_utils.mat2quat(gt44)
    d = (F.normalize(pred7[:, :4]) * F.normalize(gq[:, :4])).sum(dim=1).abs().clamp(max=1.0)
    return torch.rad2deg(2 * torch.acos(d)), (pred7[:, 4:] - gq[:, 4:]).norm(dim=1)
