"""a6/a7 — differentiable volume renderer + neural up-sampler.

Mirror of the reference's models/volume_render.py::VolRender (:11-106): same constructor, same
`forward(camera_params, feature_3d, density_3d, render_depth=False, return_origin_proj=False)`
return orders (:77-88), same `proj_origin`, same state_dict keys (`conv_rgb.{0,1,3,4,6}.*`).
PyTorch3D is not a dependency: cameras_from_opencv_projection + NDCGridRaysampler + VolumeSampler +
EmissionAbsorptionRaymarcher (+ the README.md:26-33 depth patch) are one HIP launch
(forge_render_fwd) that never materialises ray points or sampled tensors.

Differences that are deliberate:
  * `camera_params['K']` is NOT mutated in place (the reference halves the caller's tensor, :50-51);
  * optional `view2vol`: when given, `feature_3d`/`density_3d` hold each scene volume ONCE and
    view v renders volume view2vol[v] — replaces the V-fold `repeat` of models/model.py:138-139.
    Without it the reference contract (one volume per view) applies unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class VolRender(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.img_size = config.dataset.img_size
        self.volume_physical_size = config.render.volume_size
        self.n_pts_per_ray = config.render.n_pts_per_ray
        self.min_depth = config.render.min_depth
        self.max_depth = config.render.max_depth
        self.k_size = config.render.k_size
        self.pad_size = self.k_size // 2
        self.conv_rgb = nn.Sequential(
            nn.ConvTranspose2d(16, 16, kernel_size=self.k_size + 1, stride=2, padding=self.pad_size),
            nn.BatchNorm2d(16),
            nn.LeakyReLU(inplace=True),
            nn.Conv2d(16, 8, kernel_size=self.k_size, stride=1, padding=self.pad_size),
            nn.BatchNorm2d(8),
            nn.LeakyReLU(inplace=True),
            nn.Conv2d(8, 3, kernel_size=self.k_size, stride=1, padding=self.pad_size),
        )

    @staticmethod
    def _half_res_intrinsics(K):
        K = K.to(torch.float32) / 2.0          # copy; volume_render.py:50-51 without the in-place write
        K[:, -1, -1] = 1.0
        return K

    def _pack_cameras(self, camera_params, device):
        R = camera_params["R"].to(device=device, dtype=torch.float32)
        T = camera_params["T"].to(device=device, dtype=torch.float32)
        K = self._half_res_intrinsics(camera_params["K"].to(device))
        V = R.shape[0]
        cam = torch.cat([R.reshape(V, 9), T.reshape(V, 3), K[:, 0, 0:1], K[:, 1, 1:2], K[:, 0, 2:3], K[:, 1, 2:3]], dim=1)
        return cam, T, K

    @staticmethod
    def _origin_proj(T, K):
        """cameras.transform_points_screen(origin) (:77-79) == OpenCV projection of the world origin
        at half resolution."""
        return torch.stack([K[:, 0, 0] * T[:, 0] / T[:, 2] + K[:, 0, 2],
                            K[:, 1, 1] * T[:, 1] / T[:, 2] + K[:, 1, 2]], dim=-1)

    def forward(self, camera_params, feature_3d, density_3d, render_depth=False, return_origin_proj=False,
                view2vol=None):
        nvol, C, D, H, W = feature_3d.shape
        device = feature_3d.device
        cam, T, K = self._pack_cameras(camera_params, device)
        V = cam.shape[0]
        if view2vol is None:
            if V != nvol:
                raise ValueError("VolRender: %d cameras for %d volumes (pass view2vol to share volumes)" % (V, nvol))
            view2vol = torch.arange(V, dtype=torch.int32, device=device)
        else:
            view2vol = view2vol.to(device=device, dtype=torch.int32).contiguous()
        Hr = Wr = self.img_size // 2
        vox = self.volume_physical_size / D                       # :58 single_voxel_size
        half = (0.5 * (W - 1) * vox, 0.5 * (H - 1) * vox, 0.5 * (D - 1) * vox)
        outs = ops.render_rays(feature_3d, density_3d, cam, view2vol, Hr, Wr, self.n_pts_per_ray,
                               self.min_depth, self.max_depth, half, want_depth=render_depth)
        rendered_imgs = F.relu(self.conv_rgb(outs[0]))
        rendered_silhouettes = F.interpolate(outs[1], size=[self.img_size] * 2, mode="bilinear", align_corners=False)
        result = [rendered_imgs, rendered_silhouettes]
        if render_depth:
            result.append(F.interpolate(outs[2], size=[self.img_size] * 2, mode="bilinear", align_corners=False))
        if return_origin_proj:
            result.append(self._origin_proj(T, K))
        return tuple(result)

    def proj_origin(self, camera_params, device):
        """models/volume_render.py:91-103"""
        _, T, K = self._pack_cameras(camera_params, device)
        return self._origin_proj(T, K)
