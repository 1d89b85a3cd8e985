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
    """forge_rotate_fwd / forge_rotate_bwd (models/rotate.py:127-141)."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, vox, xf, mode):
        _require_cuda(vox, xf, mode)
        vox_cl = to_channels_last_3d(vox)
        n, C, D, H, W = vox_cl.shape
        xf_c = xf.detach().to(torch.float32).contiguous()
        out = _empty_like_cl(vox_cl)
        _lib.check(_lib.lib().forge_rotate_fwd(_lib.ptr(vox_cl), _lib.ptr(xf_c), _lib.ptr(mode), _lib.ptr(out),
                                               n, C, D, H, W, _lib.current_stream()), "forge_rotate_fwd")
        ctx.save_for_backward(vox_cl, xf_c, mode)
        return out

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, g):
        vox_cl, xf_c, mode = ctx.saved_tensors
        n, C, D, H, W = vox_cl.shape
        g_cl = to_channels_last_3d(g)
        dvox = _empty_like_cl(vox_cl)                      # written by the gather kernel
        dxf = torch.zeros_like(xf_c) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.lib().forge_rotate_bwd(_lib.ptr(g_cl), _lib.ptr(vox_cl), _lib.ptr(xf_c), _lib.ptr(mode),
                                               _lib.ptr(dvox), _lib.ptr(dxf), n, C, D, H, W, _lib.current_stream()),
                   "forge_rotate_bwd")
        return (dvox if ctx.needs_input_grad[0] else None), dxf, None


def rotate_warp(vox, xf, mode):
    """vox [n,C,D,H,W]; xf [n,12] 3x4 affine in normalised grid coords; mode [n] int32 (0 copy, 1 warp)."""
    return _RotateWarp.apply(vox, xf, mode)


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
        dfeat = _zeros_like_cl(feat_cl)
        ddens = torch.zeros_like(dens_c)
        dcam = torch.zeros_like(cam_c) if ctx.needs_input_grad[2] else None        # pose refinement / joint training
        _lib.check(_lib.lib().forge_render_bwd(
            _lib.ptr(feat_cl), _lib.ptr(dens_c), _lib.ptr(cam_c), _lib.ptr(view2vol),
            _lib.ptr(g_feat), _lib.ptr(g_opac), _lib.ptr(g_depth), _lib.ptr(dfeat), _lib.ptr(ddens), _lib.ptr(dcam),
            V, nvol, C, D, H, W, Hr, Wr, S, zmin, zmax, half[0], half[1], half[2], _lib.current_stream()),
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
