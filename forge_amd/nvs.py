"""Row f3 — 360-degree novel-view synthesis from one fused volume (demo.py:85-101, kubric_eval.py:166-232).

The reference samples 28 cameras with PyTorch3D's `look_at_view_transform` (elev 0, azim linspace(0,360,28)+180,
dist = render.camera_z), passes the PyTorch3D-convention (R, T) to `render()` AS IF they were OpenCV extrinsics
(a quirk that is reproduced literally: at azim = 180 deg both conventions give the canonical camera R = I,
T = (0,0,camera_z)), clamps densities to <= 1 and renders 4 chunks of 7 views, each chunk with the volume
repeated 7 times. Here all views are ONE ray-marcher launch over the single volume (view2vol = 0).
"""
import math

import torch
import torch.nn.functional as F


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, device="cpu"):
    """PyTorch3D 0.7.0 semantics for at = origin, up = +y: camera centre
    C = dist (cos(elev) sin(azim), sin(elev), cos(elev) cos(azim)); R columns = (x, y, z) camera axes in world with
    z = normalize(-C), x = normalize(up x z), y = normalize(z x x); T = -R^T C. Returns R [N,3,3], T [N,3]."""
    as_t = lambda v: torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)
    dist, elev, azim = as_t(dist), as_t(elev), as_t(azim)
    n = max(dist.numel(), elev.numel(), azim.numel())
    dist, elev, azim = dist.expand(n), elev.expand(n), azim.expand(n)
    if degrees:
        elev, azim = elev * (math.pi / 180.0), azim * (math.pi / 180.0)
    C = torch.stack([dist * torch.cos(elev) * torch.sin(azim), dist * torch.sin(elev), dist * torch.cos(elev) * torch.cos(azim)], dim=1)
    up = torch.tensor([0.0, 1.0, 0.0], device=device).expand(n, 3)
    z = F.normalize(-C, eps=1e-5)
    x = F.normalize(torch.cross(up, z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    degenerate = torch.isclose(x, torch.zeros_like(x), atol=5e-3).all(dim=1, keepdim=True)
    x = torch.where(degenerate, F.normalize(torch.cross(y, z, dim=1), eps=1e-5), x)
    R = torch.stack([x, y, z], dim=2)                       # columns = camera axes
    T = -torch.einsum("nji,nj->ni", R, C)
    return R, T


def nvs_cameras(camera_z, n_views=28, device="cpu"):
    """demo.py:84-88 / kubric_eval.py:193-196"""
    elev = torch.linspace(0, 0, n_views)
    azim = torch.linspace(0, 360, n_views) + 180
    return look_at_view_transform(dist=camera_z, elev=elev, azim=azim, device=device)


@torch.no_grad()
def render_360(model, features_mv, densities_mv, K_cv2, camera_z, n_views=28, render_depth=False):
    """features_mv [b,16,D,D,D], densities_mv [b,1,D,D,D] (outputs of the heads), K_cv2 [3,3] full-resolution intrinsics.
    Returns (imgs [b,n_views,3,H,W], masks [b,n_views,1,H,W][, depths]) — one renderer launch for all b*n_views views."""
    b = features_mv.shape[0]
    dev = features_mv.device
    R, T = nvs_cameras(camera_z, n_views, dev)
    cams = {"R": R.repeat(b, 1, 1), "T": T.repeat(b, 1), "K": K_cv2.to(dev)[None].repeat(b * n_views, 1, 1)}
    v2v = torch.arange(b, dtype=torch.int32, device=dev).repeat_interleave(n_views)
    outs = model.render(cams, features_mv, densities_mv.clamp(max=1.0), render_depth=render_depth, view2vol=v2v)
    return tuple(o.reshape(b, n_views, *o.shape[1:]) for o in outs)
