"""pytorch3d.renderer.cameras.PerspectiveCameras (NDC-space intrinsics, row-vector convention):
view = X_world @ R + T ;  ndc = (fx*X/Z + px, fy*Y/Z + py, 1/Z)."""
import torch
from ..transforms import Transform3d


class PerspectiveCameras:
    def __init__(self, R, T, focal_length, principal_point, image_size=None, device="cpu", in_ndc=True):
        self.R, self.T = R, T
        self.focal_length, self.principal_point = focal_length, principal_point
        self.image_size = image_size
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self._N = R.shape[0]
        assert in_ndc

    def to(self, device):
        mv = lambda t: t.to(device) if torch.is_tensor(t) else t
        return PerspectiveCameras(mv(self.R), mv(self.T), mv(self.focal_length), mv(self.principal_point),
                                  mv(self.image_size), device=device)

    def get_world_to_view_transform(self):
        m = torch.zeros(self._N, 4, 4, dtype=self.R.dtype, device=self.R.device)
        m[:, :3, :3] = self.R
        m[:, 3, :3] = self.T
        m[:, 3, 3] = 1.0
        return Transform3d(m)

    def get_projection_transform(self):
        K = torch.zeros(self._N, 4, 4, dtype=self.R.dtype, device=self.R.device)
        K[:, 0, 0] = self.focal_length[:, 0]
        K[:, 1, 1] = self.focal_length[:, 1]
        K[:, 0, 2] = self.principal_point[:, 0]
        K[:, 1, 2] = self.principal_point[:, 1]
        K[:, 3, 2] = 1.0
        K[:, 2, 3] = 1.0
        return Transform3d(K.transpose(1, 2).contiguous())

    def get_full_projection_transform(self):
        return self.get_world_to_view_transform().compose(self.get_projection_transform())

    def unproject_points(self, xy_depth, world_coordinates=True, from_ndc=False):
        t = self.get_full_projection_transform() if world_coordinates else self.get_projection_transform()
        inv = t.inverse()
        xy_inv_depth = torch.cat((xy_depth[..., :2], 1.0 / xy_depth[..., 2:3]), dim=-1)
        return inv.transform_points(xy_inv_depth)

    def transform_points_ndc(self, points, eps=None):
        return self.get_full_projection_transform().transform_points(points, eps=eps)

    def transform_points_screen(self, points, eps=None, with_xyflip=True):
        ndc = self.transform_points_ndc(points, eps=eps)
        image_size = self.image_size.view(-1, 2).to(ndc)
        height, width = image_size.unbind(1)
        scale = image_size.min(dim=1).values / 2.0
        K = torch.zeros(self._N, 4, 4, dtype=ndc.dtype, device=ndc.device)
        K[:, 0, 0] = scale
        K[:, 1, 1] = scale
        K[:, 0, 3] = -1.0 * width / 2.0
        K[:, 1, 3] = -1.0 * height / 2.0
        K[:, 2, 2] = 1.0
        K[:, 3, 3] = 1.0
        tr = Transform3d(K.transpose(1, 2).contiguous())
        if with_xyflip:
            flip = torch.eye(4, dtype=ndc.dtype, device=ndc.device)
            flip[0, 0] = -1.0
            flip[1, 1] = -1.0
            tr = tr.compose(Transform3d(flip[None].expand(self._N, 4, 4).transpose(1, 2).contiguous()))
        return tr.transform_points(ndc, eps=eps)
