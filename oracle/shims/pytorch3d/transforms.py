"""Row-vector 4x4 transforms, after pytorch3d.transforms.Transform3d (points @ M)."""
import torch


class Transform3d:
    def __init__(self, matrix):
        self._m = matrix  # [N,4,4], applied as p_row @ M

    def get_matrix(self):
        return self._m

    def compose(self, *others):
        m = self._m
        for o in others:
            m = m @ o.get_matrix()
        return Transform3d(m)

    def inverse(self):
        return Transform3d(torch.inverse(self._m))

    def transform_points(self, points, eps=None):
        # points [N,P,3] (or [P,3] broadcast over N)
        if points.dim() == 2:
            points = points[None]
        points = points.to(self._m.dtype)             # no-op in the fp32 runs; lets make_golden's float64 yardstick run through fp32 literals
        ones = torch.ones_like(points[..., :1])
        ph = torch.cat([points, ones], dim=-1)
        out = torch.matmul(ph, self._m)            # _broadcast_bmm
        denom = out[..., 3:]
        if eps is not None:
            sign = denom.sign() + (denom == 0.0).type_as(denom)
            denom = sign * torch.clamp(denom.abs(), eps)
        return out[..., :3] / denom
