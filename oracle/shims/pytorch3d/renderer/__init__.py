"""pytorch3d.renderer: NDCGridRaysampler, VolumeRenderer(+VolumeSampler), EmissionAbsorptionRaymarcher,
look_at_view_transform — 0.7.0 semantics incl. the reference's README.md:26-33 depth patch."""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .cameras import PerspectiveCameras  # noqa: F401


class NDCGridRaysampler(torch.nn.Module):
    def __init__(self, image_width, image_height, n_pts_per_ray, min_depth, max_depth):
        super().__init__()
        if image_width >= image_height:
            range_x, range_y = image_width / image_height, 1.0
        else:
            range_x, range_y = 1.0, image_height / image_width
        half_pix_width = range_x / image_width
        half_pix_height = range_y / image_height
        min_x, max_x = range_x - half_pix_width, -range_x + half_pix_width
        min_y, max_y = range_y - half_pix_height, -range_y + half_pix_height
        self._n_pts_per_ray, self._min_depth, self._max_depth = n_pts_per_ray, min_depth, max_depth
        Y, X = torch.meshgrid(torch.linspace(min_y, max_y, image_height, dtype=torch.float32),
                              torch.linspace(min_x, max_x, image_width, dtype=torch.float32), indexing="ij")
        self.register_buffer("_xy_grid", torch.stack([X, Y], dim=-1), persistent=False)

    def forward(self, cameras, **kwargs):
        batch_size = cameras.R.shape[0]
        device = cameras.device
        xy_grid = self._xy_grid.to(device=device, dtype=cameras.R.dtype)[None].expand(batch_size, *self._xy_grid.shape)
        spatial = xy_grid.shape[1:-1]
        n_rays = spatial[0] * spatial[1]
        depths = torch.linspace(self._min_depth, self._max_depth, self._n_pts_per_ray,
                                dtype=xy_grid.dtype, device=device)
        rays_zs = depths[None, None].expand(batch_size, n_rays, self._n_pts_per_ray)
        xy = xy_grid.reshape(batch_size, n_rays, 2)
        to_unproject = torch.cat((
            xy.view(batch_size, 1, n_rays, 2).expand(batch_size, 2, n_rays, 2).reshape(batch_size, n_rays * 2, 2),
            torch.cat((xy.new_ones(batch_size, n_rays, 1), 2.0 * xy.new_ones(batch_size, n_rays, 1)), dim=1),
        ), dim=-1)
        unprojected = cameras.unproject_points(to_unproject, from_ndc=True)
        plane1 = unprojected[:, :n_rays]
        plane2 = unprojected[:, n_rays:]
        directions = plane2 - plane1
        origins = plane1 - directions
        return SimpleNamespace(
            origins=origins.view(batch_size, *spatial, 3),
            directions=directions.view(batch_size, *spatial, 3),
            lengths=rays_zs.reshape(batch_size, *spatial, self._n_pts_per_ray),
            xys=xy_grid,
        )


class EmissionAbsorptionRaymarcher(torch.nn.Module):
    def __init__(self, surface_thickness=1):
        super().__init__()
        self.surface_thickness = surface_thickness

    def forward(self, rays_densities, rays_features, eps=1e-10, **kwargs):
        rays_densities = rays_densities[..., 0]
        x = (1.0 + eps) - rays_densities
        cp = torch.cumprod(x, dim=-1)
        s = self.surface_thickness
        absorption = torch.cat([torch.ones_like(cp[..., :s]), cp[..., :-s]], dim=-1)
        weights = rays_densities * absorption
        features = (weights[..., None] * rays_features).sum(dim=-2)
        opacities = 1.0 - torch.prod(1.0 - rays_densities, dim=-1, keepdim=True)
        # --- reference README.md:26-33 patch ---
        if 'render_depth' in kwargs.keys() and kwargs['render_depth'] == True and 'ray_bundle' in kwargs.keys():
            ray_bundle_lengths = kwargs['ray_bundle'].lengths[..., None]
            depths = (weights[..., None] * ray_bundle_lengths).sum(dim=-2)
            return torch.cat((features, opacities, depths), dim=-1)
        return torch.cat((features, opacities), dim=-1)


class VolumeSampler(torch.nn.Module):
    def __init__(self, volumes, sample_mode="bilinear"):
        super().__init__()
        self._volumes, self._sample_mode = volumes, sample_mode

    def forward(self, ray_bundle, **kwargs):
        pts_world = ray_bundle.origins[..., None, :] + ray_bundle.lengths[..., :, None] * ray_bundle.directions[..., None, :]
        pts_local = self._volumes.world_to_local_coords(pts_world)
        b = pts_local.shape[0]
        flat = pts_local.view(b, -1, 1, 1, 3)
        dens = F.grid_sample(self._volumes.densities(), flat, align_corners=True, mode=self._sample_mode)
        dens = dens.permute(0, 2, 3, 4, 1).reshape(*pts_local.shape[:-1], dens.shape[1])
        feat = self._volumes.features()
        if feat is not None:
            feat = F.grid_sample(feat, flat, align_corners=True, mode=self._sample_mode)
            feat = feat.permute(0, 2, 3, 4, 1).reshape(*pts_local.shape[:-1], feat.shape[1])
        return dens, feat


class VolumeRenderer(torch.nn.Module):
    def __init__(self, raysampler, raymarcher, sample_mode="bilinear"):
        super().__init__()
        self.raysampler, self.raymarcher, self._sample_mode = raysampler, raymarcher, sample_mode

    def forward(self, cameras, volumes, **kwargs):
        ray_bundle = self.raysampler(cameras=cameras, volumetric_function=None, **kwargs)
        dens, feat = VolumeSampler(volumes, self._sample_mode)(ray_bundle, cameras=cameras, **kwargs)
        images = self.raymarcher(rays_densities=dens, rays_features=feat, ray_bundle=ray_bundle, **kwargs)
        return images, ray_bundle


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, at=((0, 0, 0),), up=((0, 1, 0),), device="cpu"):
    """PyTorch3D-convention R [N,3,3] (row-vector, columns = camera x,y,z axes in world), T [N,3]."""
    t = lambda v: torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)
    dist, elev, azim = t(dist), t(elev), t(azim)
    n = max(dist.numel(), elev.numel(), azim.numel())
    dist, elev, azim = dist.expand(n), elev.expand(n), azim.expand(n)
    if degrees:
        elev, azim = elev * math.pi / 180.0, azim * math.pi / 180.0
    x = dist * torch.cos(elev) * torch.sin(azim)
    y = dist * torch.sin(elev)
    z = dist * torch.cos(elev) * torch.cos(azim)
    at_t = torch.as_tensor(at, dtype=torch.float32, device=device).expand(n, 3)
    up_t = torch.as_tensor(up, dtype=torch.float32, device=device).expand(n, 3)
    C = torch.stack([x, y, z], dim=1) + at_t
    z_axis = F.normalize(at_t - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(up_t, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        x_axis = torch.where(is_close, F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5), x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T
