"""a8 — FORGE_poseEstimator3D: the model class driven by kubric_train_pose_3D.py (GT-pose training,
BASELINE config 4).

Mirror of the reference's models/model_single_pose_estimator.py (:14-138): same constructor, same
sub-module attribute names (`encoder_3d`, `render`, `rotate`, `encoder_traj`), same
`forward(sample, dataset, device)` and return tuples per mode (:135-138, :89-99).

MI355X-first differences (results identical, memory traffic not):
  * the three fused volumes (2-view, 3-view, all-view) are rendered by ONE renderer launch through a
    view->volume index instead of materialising 2t repeated copies of the 17-channel 64^3 volumes
    (reference :110-129 builds [b*2t,16,64^3] + [b*2t,1,64^3] with repeat/cat);
  * the two heads run once on the concatenated [3b] fused volumes;
  * K is never modified in place.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoder import Encoder3D
from .pose_estimator_3d import PoseEstimator3D
from .rotate import Rotate_world
from .volume_render import VolRender


class FORGE_poseEstimator3D(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder_3d = Encoder3D(config)
        self.render = VolRender(config)
        self.rotate = Rotate_world(config)
        self.encoder_traj = PoseEstimator3D(config)

    def forward(self, sample, dataset, device):
        clips = sample["images"].to(device)
        b, t, c, h, w = clips.shape
        features_raw = self.encoder_3d.get_feat3D(clips.reshape(b * t, c, h, w))      # [b*t,C,D,H,W]
        _, C, D, H, W = features_raw.shape
        features_raw = features_raw.reshape(b, t, C, D, H, W)

        if not self.config.train.use_gt_pose:
            poses_cam, conf = self.encoder_traj(features_raw)                          # :45
            tmp = torch.zeros_like(poses_cam)
            tmp[:, :4] = F.normalize(poses_cam[:, :4])
            tmp[:, 4:] = poses_cam[:, 4:]
            poses_cam = tmp
            camPoseRel_cv2 = self.encoder_traj.toSE3(poses_cam)                       # [b*(t-1),4,4]
            canonical_pose_cv2 = dataset.get_canonical_pose_cv2(device=device)
            canonical_extrinsics_cv2 = dataset.get_canonical_extrinsics_cv2(device=device)
            camPoses_cv2 = canonical_pose_cv2.unsqueeze(0) @ camPoseRel_cv2
            camE_cv2 = torch.inverse(camPoses_cv2).reshape(b, t - 1, 4, 4)
            camPoses_cv2 = camPoses_cv2.reshape(b, t - 1, 4, 4)
            camPoses_cv2 = torch.cat([canonical_pose_cv2.reshape(1, 1, 4, 4).repeat(b, 1, 1, 1), camPoses_cv2], dim=1)
            camE_cv2 = torch.cat([canonical_extrinsics_cv2.reshape(1, 1, 4, 4).repeat(b, 1, 1, 1), camE_cv2], dim=1)
            from .geo_utils import mat2quat
            poses_cam_gt = mat2quat(sample["cam_poses_rel_cv2"][:, 1:].to(device).reshape(b * (t - 1), 4, 4))
            camPose_return = {"gt": poses_cam_gt, "pred": poses_cam, "conf": conf}
        else:
            if self.config.train.canonicalize:
                camE_cv2 = sample["cam_extrinsics_cv2_canonicalized"].to(device)
                camPoses_cv2 = sample["cam_poses_cv2_canonicalized"].to(device)
            else:
                camE_cv2 = sample["cam_extrinsics_cv2"].to(device)
                camPoses_cv2 = sample["cam_poses_cv2"].to(device)
            camPose_return = None

        # cameras for rendering: every input camera twice (:77-85)
        camE_cv2 = camE_cv2.repeat(1, 2, 1, 1)
        camPoses_cv2 = camPoses_cv2.repeat(1, 2, 1, 1)
        camK = sample["K_cv2"].repeat(1, 2, 1, 1).to(device)
        cameras = {
            "R": camE_cv2.reshape(b * 2 * t, 4, 4)[:, :3, :3],
            "T": camE_cv2.reshape(b * 2 * t, 4, 4)[:, :3, 3],
            "K": camK.reshape(b * 2 * t, 3, 3),
        }

        if self.config.train.parameter == "pose":                                      # :87-99
            origin_proj = self.render.proj_origin(cameras, device)
            return camPose_return, 2 * origin_proj / self.config.dataset.img_size

        features_transformed = self.rotate(voxels=features_raw, camPoses_cv2=camPoses_cv2[:, :t], grid_size=D)

        # three fusions: first 3 views, last 2 views, all views (:108-109, :120)
        features_3v, features_2v, features_mv = self.encoder_3d.fuse_groups(
            features_transformed, [list(range(min(3, t))), list(range(max(t - 2, 0), t)), list(range(t))])
        if self.encoder_3d.training:
            # BatchNorm batch statistics (and the running-stat updates) follow the reference's call structure: the heads run on
            # cat([3v, 2v]) (:110-111) and on the all-view volume (:121-122) SEPARATELY - one 3b batch would normalise differently
            f32 = torch.cat([features_3v, features_2v], dim=0)
            d32, r32 = self.encoder_3d.get_density3D(f32), self.encoder_3d.get_render_features(f32)
            dm, rm = self.encoder_3d.get_density3D(features_mv), self.encoder_3d.get_render_features(features_mv)
            densities, features = torch.cat([d32, dm], dim=0), torch.cat([r32, rm], dim=0)
        else:
            fused = torch.cat([features_3v, features_2v, features_mv], dim=0)          # eval BN: one [3b,128,D,H,W] batch is the same arithmetic
            densities = self.encoder_3d.get_density3D(fused)                           # [3b,1,2D,..]
            features = self.encoder_3d.get_render_features(fused)                      # [3b,16,2D,..]
        if self.config.dataset.name == "omniobject3d":
            densities = densities.clamp(min=0.0, max=1.0)

        # view order per scene (:112-129): 2v volume x3 cams, 3v volume x2 cams, mv volume x t cams
        scene = torch.arange(b, device=device, dtype=torch.int32)[:, None]
        per_scene = torch.cat([(b + scene).expand(b, 3), scene.expand(b, 2), (2 * b + scene).expand(b, t)], dim=1)
        view2vol = per_scene.reshape(b * 2 * t).contiguous()

        rendered_imgs, rendered_masks, origin_proj = self.render(cameras, features, densities,
                                                                 return_origin_proj=True, view2vol=view2vol)
        if self.config.train.use_gt_pose:
            return rendered_imgs, rendered_masks
        return rendered_imgs, rendered_masks, 2 * origin_proj / self.config.dataset.img_size, camPose_return
