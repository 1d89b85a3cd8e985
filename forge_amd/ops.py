"""torch.autograd bindings of the HIP kernels in libforge_hip.so.

Host-side plumbing only: layout checks, output allocation, stream hand-off, autograd wiring.
All arithmetic of the hot path happens in forge_amd/csrc/*.hip. There is no CPU implementation
here: calling these ops with CPU tensors (or without the built library) raises.
"""
import torch

from . import _lib


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("forge_amd ops need tensors on the MI355X (cuda/HIP device); got a %s tensor. "
                               "There is no CPU fallback." % t.device)


def to_channels_last_3d(x):
    """[N,C,D,H,W] float32 -> same logical tensor whose memory is [N,D,H,W,C] (no copy if it already is)."""
    if x.dtype != torch.float32:
        raise TypeError("forge_amd ops are fp32 (got %s)" % x.dtype)
    if x.permute(0, 2, 3, 4, 1).is_contiguous():
        return x
    return x.contiguous(memory_format=torch.channels_last_3d)


def _empty_like_cl(x_cl):
    n, c, d, h, w = x_cl.shape
    return torch.empty((n, d, h, w, c), dtype=x_cl.dtype, device=x_cl.device).permute(0, 4, 1, 2, 3)


def _zeros_like_cl(x_cl):
    n, c, d, h, w = x_cl.shape
    return torch.zeros((n, d, h, w, c), dtype=x_cl.dtype, device=x_cl.device).permute(0, 4, 1, 2, 3)


class _RotateWarp(torch.autograd.Function):
    """forge_rotate_fwd / forge_rotate_bwd (models/rotate.py:127-141). slot (optional, int32 [n]): the warp stores view i at volume slot[i]
    (the view order of models/model.py:127-128 fused into the store, forge_rotate_fwd_slots) and the backward reads its gradient there."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, vox, xf, mode, slot=None):
        _require_cuda(vox, xf, mode)
        vox_cl = to_channels_last_3d(vox)
        n, C, D, H, W = vox_cl.shape
        xf_c = xf.detach().to(torch.float32).contiguous()
        out = _empty_like_cl(vox_cl)
        if slot is None:
            _lib.check(_lib.lib().forge_rotate_fwd(_lib.ptr(vox_cl), _lib.ptr(xf_c), _lib.ptr(mode), _lib.ptr(out),
                                                   n, C, D, H, W, _lib.current_stream()), "forge_rotate_fwd")
        else:
            _lib.check(_lib.lib().forge_rotate_fwd_slots(_lib.ptr(vox_cl), _lib.ptr(xf_c), _lib.ptr(mode), _lib.ptr(slot), _lib.ptr(out),
                                                         n, C, D, H, W, _lib.current_stream()), "forge_rotate_fwd_slots")
        ctx.save_for_backward(vox_cl, xf_c, mode, slot)
        return out

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, g):
        vox_cl, xf_c, mode, slot = ctx.saved_tensors
        n, C, D, H, W = vox_cl.shape
        dvox = _empty_like_cl(vox_cl) if ctx.needs_input_grad[0] else None     # written by the gather kernel; frozen volumes (refinement): skipped
        dxf = torch.zeros_like(xf_c) if ctx.needs_input_grad[1] else None
        if dvox is None and dxf is None:
            return None, None, None, None
        g_cl = to_channels_last_3d(g)
        if slot is None:
            _lib.check(_lib.lib().forge_rotate_bwd(_lib.ptr(g_cl), _lib.ptr(vox_cl), _lib.ptr(xf_c), _lib.ptr(mode),
                                                   _lib.ptr(dvox), _lib.ptr(dxf), n, C, D, H, W, _lib.current_stream()), "forge_rotate_bwd")
        else:
            _lib.check(_lib.lib().forge_rotate_bwd_slots(_lib.ptr(g_cl), _lib.ptr(vox_cl), _lib.ptr(xf_c), _lib.ptr(mode), _lib.ptr(slot),
                                                         _lib.ptr(dvox), _lib.ptr(dxf), n, C, D, H, W, _lib.current_stream()), "forge_rotate_bwd_slots")
        return dvox, dxf, None, None


class _PoseChain(torch.autograd.Function):
    """forge_pose_chain_fwd / _bwd: the refinement loop's pose algebra (normalise, quaternion -> matrix, canonical @ rel, inverse, P_0 @ inverse,
    camera packing) as one launch forward (values + Jacobian by forward-mode differentiation) and one launch backward."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, rot, trans, can_p, can_e, K, half_extent, b, t):
        _require_cuda(rot, trans, can_p, can_e, K)
        dev = rot.device
        f32 = lambda x: x.detach().to(torch.float32).contiguous()
        rot_c, trans_c, Kc = f32(rot), f32(trans), f32(K).reshape(b * t, 9)
        if rot_c.shape != (b * (t - 1), 4) or trans_c.shape != (b * (t - 1), 3):
            raise ValueError("pose_chain: rot %s / trans %s do not match b=%d, t=%d" % (tuple(rot.shape), tuple(trans.shape), b, t))
        new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        xf, cam, poses, origin, jac = new(b * t, 12), new(b * t, 16), new(b, t, 4, 4), new(b * t, 2), new(b * (t - 1), 24, 7)
        mode, slot = torch.empty(b * t, dtype=torch.int32, device=dev), torch.empty(b * t, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().forge_pose_chain_fwd(_lib.ptr(rot_c), _lib.ptr(trans_c), _lib.ptr(f32(can_p)), _lib.ptr(f32(can_e)), _lib.ptr(Kc), float(half_extent),
                                                   b, t, _lib.ptr(xf), _lib.ptr(mode), _lib.ptr(slot), _lib.ptr(cam), _lib.ptr(poses), _lib.ptr(origin),
                                                   _lib.ptr(jac), _lib.current_stream()), "forge_pose_chain_fwd")
        ctx.save_for_backward(jac)
        ctx.bt = (b, t)
        ctx.mark_non_differentiable(mode, slot, poses, origin)
        return xf, cam, mode, slot, poses, origin

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dxf, dcam, _dmode, _dslot, _dposes, _dorigin):
        (jac,) = ctx.saved_tensors
        b, t = ctx.bt
        if dxf is None and dcam is None:
            return (None,) * 8
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        dxf, dcam = c(dxf), c(dcam)
        drot = torch.empty(b * (t - 1), 4, dtype=torch.float32, device=jac.device)
        dtrans = torch.empty(b * (t - 1), 3, dtype=torch.float32, device=jac.device)
        _lib.check(_lib.lib().forge_pose_chain_bwd(_lib.ptr(jac), _lib.ptr(dxf), _lib.ptr(dcam), _lib.ptr(drot), _lib.ptr(dtrans), b, t, _lib.current_stream()),
                   "forge_pose_chain_bwd")
        return drot, dtrans, None, None, None, None, None, None


def pose_chain(rot, trans, can_p, can_e, K, half_extent, b, t):
    """rot [b(t-1),4] raw quaternions, trans [b(t-1),3], can_p / can_e [4,4], K [b,t,3,3] -> (xf [b t,12], cam [b t,16], mode [b t] int32,
    slot [b t] int32 (view i goes to volume slot[i]: the order of models/model.py:152-158), poses [b,t,4,4], origin [b t,2]); gradients reach
    rot / trans through xf (the warp) and cam (the ray-marcher)."""
    return _PoseChain.apply(rot, trans, can_p, can_e, K, half_extent, b, t)


def rotate_warp(vox, xf, mode, slot=None):
    """vox [n,C,D,H,W]; xf [n,12] 3x4 affine in normalised grid coords; mode [n] int32 (0 copy, 1 warp); slot [n] int32 (optional): view i is
    stored at volume slot[i]."""
    return _RotateWarp.apply(vox, xf, mode, slot)


class _RenderRays(torch.autograd.Function):
    """forge_render_fwd / forge_render_bwd (models/volume_render.py:53-63)."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, feat, dens, cam, view2vol, Hr, Wr, S, zmin, zmax, half, want_depth):
        _require_cuda(feat, dens, cam, view2vol)
        feat_cl = to_channels_last_3d(feat)
        nvol, C, D, H, W = feat_cl.shape
        if dens.shape != (nvol, 1, D, H, W):
            raise ValueError("density volume must be [%d,1,%d,%d,%d], got %s" % (nvol, D, H, W, tuple(dens.shape)))
        dens_c = dens.to(torch.float32).contiguous()
        cam_c = cam.detach().to(torch.float32).contiguous()
        V = cam_c.shape[0]
        out_feat = torch.empty((V, Hr, Wr, C), dtype=torch.float32, device=feat.device).permute(0, 3, 1, 2)   # NCHW view of NHWC memory
        out_opac = torch.empty((V, 1, Hr, Wr), dtype=torch.float32, device=feat.device)
        out_depth = torch.empty((V, 1, Hr, Wr), dtype=torch.float32, device=feat.device) if want_depth else None
        _lib.check(_lib.lib().forge_render_fwd(
            _lib.ptr(feat_cl), _lib.ptr(dens_c), _lib.ptr(cam_c), _lib.ptr(view2vol),
            _lib.ptr(out_feat), _lib.ptr(out_opac), _lib.ptr(out_depth),
            V, nvol, C, D, H, W, Hr, Wr, S, zmin, zmax, half[0], half[1], half[2], _lib.current_stream()),
            "forge_render_fwd")
        ctx.save_for_backward(feat_cl, dens_c, cam_c, view2vol)
        ctx.cfg = (Hr, Wr, S, zmin, zmax, half, want_depth)
        if want_depth:
            return out_feat, out_opac, out_depth
        return out_feat, out_opac

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, g_feat, g_opac, g_depth=None):
        feat_cl, dens_c, cam_c, view2vol = ctx.saved_tensors
        Hr, Wr, S, zmin, zmax, half, want_depth = ctx.cfg
        nvol, C, D, H, W = feat_cl.shape
        V = cam_c.shape[0]
        g_feat = g_feat.contiguous(memory_format=torch.channels_last)      # [V,Hr,Wr,C] in memory
        g_opac = g_opac.contiguous()
        g_depth = g_depth.contiguous() if (want_depth and g_depth is not None) else None
        # every element of dfeat / ddens / dcam is WRITTEN by the voxel-parallel gather (deterministic, no atomics): no zero-fills
        dfeat = _empty_like_cl(feat_cl)
        ddens = torch.empty_like(dens_c)
        dcam = torch.empty_like(cam_c) if ctx.needs_input_grad[2] else None        # pose refinement / joint training
        L = _lib.lib()
        ws_bytes = L.forge_render_bwd_ws_bytes(V, C, Hr, Wr, S, 0 if dcam is None else 1)
        if ws_bytes < 0:
            raise RuntimeError("forge_amd: forge_render_bwd_ws_bytes rejected V=%d C=%d Hr=%d Wr=%d S=%d" % (V, C, Hr, Wr, S))
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=feat_cl.device)   # per-sample (dL/dd_s, T_s d_s) of every ray: 8 V Hr Wr S bytes
        _lib.check(L.forge_render_bwd(
            _lib.ptr(feat_cl), _lib.ptr(dens_c), _lib.ptr(cam_c), _lib.ptr(view2vol),
            _lib.ptr(g_feat), _lib.ptr(g_opac), _lib.ptr(g_depth), _lib.ptr(dfeat), _lib.ptr(ddens), _lib.ptr(dcam),
            V, nvol, C, D, H, W, Hr, Wr, S, zmin, zmax, half[0], half[1], half[2], _lib.ptr(ws), ws_bytes, _lib.current_stream()),
            "forge_render_bwd")
        return (dfeat, ddens, dcam) + (None,) * 8


def render_rays(feat, dens, cam, view2vol, Hr, Wr, S, zmin, zmax, half, want_depth=False):
    """feat [nvol,C,D,H,W], dens [nvol,1,D,H,W], cam [V,16] (R9,T3,fx,fy,cx,cy at half res),
    view2vol [V] int32 -> (feat [V,C,Hr,Wr], opacity [V,1,Hr,Wr][, depth [V,1,Hr,Wr]])."""
    return _RenderRays.apply(feat, dens, cam, view2vol, int(Hr), int(Wr), int(S), float(zmin), float(zmax),
                             tuple(float(h) for h in half), bool(want_depth))


class _ResizeBilinear(torch.autograd.Function):
    """forge_resize_bilinear_fwd / _bwd: planes [..., Hi, Wi] -> [..., Ho, Wo], bilinear, align_corners=False (models/volume_render.py:69,74)."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, Ho, Wo):
        _require_cuda(x)
        if x.dtype != torch.float32:
            raise TypeError("forge_amd ops are fp32 (got %s)" % x.dtype)
        xc = x.contiguous()
        Hi, Wi = xc.shape[-2:]
        P = xc.numel() // (Hi * Wi)
        out = torch.empty(xc.shape[:-2] + (Ho, Wo), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().forge_resize_bilinear_fwd(_lib.ptr(xc), _lib.ptr(out), P, Hi, Wi, Ho, Wo, _lib.current_stream()), "forge_resize_bilinear_fwd")
        ctx.dims = (P, Hi, Wi, Ho, Wo, tuple(x.shape))
        return out

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, g):
        P, Hi, Wi, Ho, Wo, shape = ctx.dims
        gc = g.contiguous()
        din = torch.empty(shape, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().forge_resize_bilinear_bwd(_lib.ptr(gc), _lib.ptr(din), P, Hi, Wi, Ho, Wo, _lib.current_stream()), "forge_resize_bilinear_bwd")
        return din, None, None


def resize_bilinear(x, Ho, Wo):
    """F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False) on the HIP kernels (forward and adjoint)."""
    return _ResizeBilinear.apply(x, int(Ho), int(Wo))


def attention_applies(q, k, v):
    """forge_attention_fwd's domain: fp32 on the MI355X, no autograd graph wanted, one head of 64 channels, token counts multiples of 64,
    q / k [B,N,64] and v [B,Nk,64] or [1,Nk,64] (shared)."""
    return (q.is_cuda and k.device == q.device and v.device == q.device                 # raw pointers go to the kernel: all three on q's HIP device
            and q.dtype == torch.float32 and k.dtype == torch.float32 and v.dtype == torch.float32 and not torch.is_grad_enabled()
            and q.dim() == 3 and k.dim() == 3 and v.dim() == 3 and q.shape[-1] == 64 and k.shape[-1] == 64 and v.shape[-1] == 64
            and q.shape[0] == k.shape[0] and v.shape[0] in (1, q.shape[0]) and v.shape[1] == k.shape[1]
            and q.shape[1] % 64 == 0 and k.shape[1] % 64 == 0 and q.shape[1] > 0 and k.shape[1] > 0)


@_lib.on_tensor_device
def attention(q, k, v):
    """softmax(q k^T) v for one head (models/model_utils.py:207-229, unscaled) without materialising the [B,Nq,Nk] matrix: forge_attention_fwd.
    q [B,Nq,64], k [B,Nk,64], v [B,Nk,64] or [1,Nk,64] (one value table for every batch element) -> [B,Nq,64]. Inference only (no autograd node)."""
    if not attention_applies(q, k, v):
        raise RuntimeError("forge_amd: ops.attention needs fp32 [B,N,64] tensors on the MI355X with token counts that are multiples of 64, outside autograd "
                           "(got q %s, k %s, v %s, grad mode %s)" % (tuple(q.shape), tuple(k.shape), tuple(v.shape), torch.is_grad_enabled()))
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, Nq, d = q.shape
    Nk = k.shape[1]
    out = torch.empty(B, Nq, d, dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().forge_attention_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), 0 if v.shape[0] == 1 and B > 1 else Nk, _lib.ptr(out), B, Nq, Nk, d,
                                              _lib.current_stream()), "forge_attention_fwd")
    return out
