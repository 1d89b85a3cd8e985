"""pytorch3d.utils.camera_conversions.cameras_from_opencv_projection (0.7.0 semantics)."""
import torch
from ..renderer.cameras import PerspectiveCameras


def cameras_from_opencv_projection(R, tvec, camera_matrix, image_size):
    focal_length = torch.stack([camera_matrix[:, 0, 0], camera_matrix[:, 1, 1]], dim=-1)
    principal_point = camera_matrix[:, :2, 2]
    image_size_wh = image_size.to(R).flip(dims=(1,))
    scale = image_size_wh.min(dim=1, keepdim=True)[0] / 2.0
    scale = scale.expand(-1, 2)
    c0 = image_size_wh / 2.0
    focal_p3d = focal_length / scale
    p0_p3d = -(principal_point - c0) / scale
    R_p3d = R.clone().permute(0, 2, 1)
    T_p3d = tvec.clone()
    R_p3d[:, :, :2] *= -1
    T_p3d[:, :2] *= -1
    return PerspectiveCameras(R=R_p3d, T=T_p3d, focal_length=focal_p3d,
                              principal_point=p0_p3d, image_size=image_size, device=R.device)
