"""a2 — voxel-grid pose warp.

Mirror of the reference's models/rotate.py::Rotate_world (:9-156): same constructor / forward
signature (`forward(voxels, camPoses_cv2, grid_size=32)`) and the same — unused but checkpointed —
parameters `conv3d_1..4` (:37-45). The warp itself (affine grid + trilinear grid_sample with
zeros padding and align_corners=False, then cat with view 0; :125-141) is ONE HIP launch
(forge_rotate_fwd); the 4x4 algebra T = P_0 P_i^-1 (:64-89) stays in torch so that gradients
reach predicted poses through autograd.
"""
import torch
import torch.nn as nn

from . import _lib, geo_utils, ops

_SUPPORTED = (16, 32, 48, 64, 128)   # grids the reference pre-computes (models/rotate.py:18-35)


class Rotate_world(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.padding_mode = config.network.padding_mode      # read but ignored, as in the reference (:138)
        self.grid_size = 32
        self.vol_size = config.render.volume_size
        self.single_voxel_size = self.vol_size / self.grid_size
        self.grid_coord_max = self.half_extent(32)           # "should be 0.4844" (:23)
        self.conv3d_1 = nn.Conv3d(16, 16, 3, padding=1)
        self.conv3d_2 = nn.Conv3d(16, 16, 3, padding=1)
        self.conv3d_3 = nn.Conv3d(128, 128, 3, padding=1)
        self.conv3d_4 = nn.Conv3d(128, 128, 3, padding=1)
        for m in (self.conv3d_1, self.conv3d_2, self.conv3d_3, self.conv3d_4):   # train_utils.normal_init
            nn.init.normal_(m.weight, 0.0, 0.01)
            nn.init.constant_(m.bias, 0)

    def half_extent(self, grid_size):
        """pytorch3d Volumes.get_coord_grid(world_coordinates=True).max() = 0.5 (D-1) vol/D (:22-35)"""
        return 0.5 * (grid_size - 1) * (self.vol_size / grid_size)

    def get_transformation(self, camPoses_cv2):
        """models/rotate.py:64-89: T = P_0 @ inverse(P_i), i = 1..t-1 -> [B*(t-1),4,4]"""
        B, t = camPoses_cv2.shape[:2]
        pose_0 = camPoses_cv2[:, 0:1].repeat(1, t - 1, 1, 1).reshape(B * (t - 1), 4, 4)
        pose_1 = camPoses_cv2[:, 1:].reshape(B * (t - 1), 4, 4)
        return pose_0 @ geo_utils.inverse_affine(pose_1)        # poses are affine (last row 0 0 0 1): closed form, no host sync

    @_lib.on_tensor_device
    def forward(self, voxels, camPoses_cv2, grid_size=32, order=None):
        """voxels [B,t,C,D,H,W], camPoses_cv2 [B,t,4,4] -> [B,t,C,D,H,W] (view 0 unchanged).
        order (extension): the view permutation of models/model.py:127-128 applied to the output, out[:, j] = warped[:, order[:, j]] -
        either [B,t] indices or the string "distance" (= sequence_from_distance of the poses' camera positions). In inference the
        kernel's store applies it (forge_rotate_fwd_slots; for "distance" the ranks are computed by the pose kernel: no sort / gather
        launches at all); under autograd it is a gather on the result. The reference's callers never pass it."""
        B, t, C, D, H, W = voxels.shape
        if grid_size not in _SUPPORTED:
            raise ValueError("Rotate_world: grid_size %r not in %s (models/rotate.py:109-123)" % (grid_size, _SUPPORTED))
        if not (D == H == W == grid_size):
            raise ValueError("Rotate_world: voxels %s do not match grid_size=%d" % ((D, H, W), grid_size))
        device = voxels.device
        e = self.half_extent(grid_size)
        poses = camPoses_cv2.to(device=device, dtype=torch.float32)
        if not (torch.is_grad_enabled() and poses.requires_grad):
            # no gradient to the poses: T = P_0 P_i^-1 and the affine packing run in one tiny kernel (no torch.inverse,
            # whose LU + info check costs more host time than the whole warp)
            xf = torch.empty(B * t, 12, dtype=torch.float32, device=device)
            mode = torch.empty(B * t, dtype=torch.int32, device=device)
            fused_order = order is not None and not voxels.requires_grad
            by_distance = fused_order and isinstance(order, str)
            slot = torch.empty(B * t, dtype=torch.int32, device=device) if by_distance else None
            trans = poses[:, :, :3, 3]                                                      # keys exactly as sequence_from_distance computes them
            dist = ((trans - trans[:, 0:1]) ** 2).sum(dim=-1).contiguous() if by_distance else None
            _lib.check(_lib.lib().forge_rotate_xf_from_poses(_lib.ptr(poses.contiguous()), _lib.ptr(xf), _lib.ptr(mode), _lib.ptr(slot), _lib.ptr(dist),
                                                             B, t, e, _lib.current_stream()), "forge_rotate_xf_from_poses")
            if fused_order:
                if not by_distance:
                    inv = torch.argsort(order.to(device), dim=1)                            # slot of view i in the ordered stack
                    slot = (inv + torch.arange(B, device=device)[:, None] * t).to(torch.int32).reshape(B * t).contiguous()
                vox_cl = ops.to_channels_last_3d(voxels.reshape(B * t, C, D, H, W))
                out = ops._empty_like_cl(vox_cl)
                _lib.check(_lib.lib().forge_rotate_fwd_slots(_lib.ptr(vox_cl), _lib.ptr(xf), _lib.ptr(mode), _lib.ptr(slot), _lib.ptr(out),
                                                             B * t, C, D, H, W, _lib.current_stream()), "forge_rotate_fwd_slots")
                return out.reshape(B, t, C, D, H, W)
            out = ops.rotate_warp(voxels.reshape(B * t, C, D, H, W), xf, mode)
            return self._gather_order(out.reshape(B, t, C, D, H, W), order, poses)
        if t > 1:
            T = self.get_transformation(poses)                                                  # [B(t-1),4,4]
            xf_w = torch.cat([T[:, :3, :3], T[:, :3, 3:4] / e], dim=-1).reshape(B, t - 1, 12)
            ident = torch.zeros(B, 1, 12, dtype=torch.float32, device=device)
            xf = torch.cat([ident, xf_w], dim=1).reshape(B * t, 12)
        else:
            xf = torch.zeros(B, 12, dtype=torch.float32, device=device)
        mode = torch.ones(B, t, dtype=torch.int32, device=device)
        mode[:, 0] = 0
        out = ops.rotate_warp(voxels.reshape(B * t, C, D, H, W), xf, mode.reshape(B * t)).reshape(B, t, C, D, H, W)
        return self._gather_order(out, order, poses)

    @staticmethod
    def _gather_order(out, order, poses):
        if order is None:
            return out
        if isinstance(order, str):
            trans = poses[:, :, :3, 3].detach()
            order = torch.sort(((trans - trans[:, 0:1]) ** 2).sum(dim=-1), descending=False, stable=True)[1]
        return out[torch.arange(out.shape[0], device=out.device)[:, None], order.to(out.device)]
