"""a8 — FORGE: the full model class used by kubric_train_joint.py / kubric_eval.py / demo.py.

Mirror of the reference's models/model.py (:18-168): same constructor, sub-module attribute names
(`encoder_3d`, `render`, `rotate`, `encoder_traj`, `encoder_traj_2d`, `pose_head`), the module-level
helpers `sequence_from_distance` / `chose_selected` (imported by demo.py:21, kubric_eval.py:28) and
`forward(sample, dataset, device)` with the reference's return tuples per mode (:98-114, :145-148).

Deliberate differences:
  * `use_gt_pose=True` WORKS here. In the reference that branch is broken (`idxs`
    is never defined and 10+5 extrinsics are reshaped to 10). Here the GT branch takes the first 5
    cameras as input views, orders them by distance like the predicted branch, and appends the
    remaining cameras of the sample (if any) as novel views.
  * the fused volume is rendered for all V cameras through a view->volume index; it is never
    repeated V times (reference :138-139).
  * K is not modified in place.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import geo_utils
from .encoder import Encoder3D
from .pose_estimator_2d import PoseEstimator2D
from .pose_estimator_3d import PoseEstimator3D
from .rotate import Rotate_world
from .volume_render import VolRender


def sequence_from_distance(trans):
    """models/model.py:152-158 — translations [b,t,3] -> view order by squared distance to view 0."""
    dist = ((trans - trans[:, 0:1, :]) ** 2).sum(dim=-1)
    return torch.sort(dist, descending=False)[1]


def chose_selected(tensor, idxs):
    """models/model.py:161-168 — per-sample gather along dim 1 (one indexing op, no python loop)."""
    assert tensor.shape[0] == len(idxs)
    b = tensor.shape[0]
    return tensor[torch.arange(b, device=tensor.device)[:, None], idxs.to(tensor.device)]


class FORGE(nn.Module):
    N_INPUT = 5     # the reference hard-codes 5 input views (models/model.py:50, :83, :100)

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder_3d = Encoder3D(config)
        self.render = VolRender(config)
        self.rotate = Rotate_world(config)
        self.encoder_traj = PoseEstimator3D(config)
        self.encoder_traj_2d = PoseEstimator2D()
        self.pose_head = nn.Sequential(
            nn.Dropout(p=0.5),
            nn.Linear(2048, 512),
            nn.LayerNorm(512),
            nn.LeakyReLU(),
            nn.Linear(512, self.encoder_traj.pose_dim + 1),
        )

    def forward(self, sample, dataset, device):
        b, t_all = sample["images"].shape[:2]
        clips = sample["images"][:, :self.N_INPUT].to(device)
        b, t, c, h, w = clips.shape
        features_raw = self.encoder_3d.get_feat3D(clips.reshape(b * t, c, h, w))
        _, C, D, H, W = features_raw.shape
        features_raw = features_raw.reshape(b, t, C, D, H, W)

        if not self.config.train.use_gt_pose:
            pose_feat_3d = self.encoder_traj(features_raw, return_features=True)       # [b(t-1),1024]
            pose_feat_2d = self.encoder_traj_2d(clips, return_features=True)          # [b(t-1),1024]
            pred = self.pose_head(torch.cat([pose_feat_3d, pose_feat_2d], dim=-1))
            poses_cam, conf = pred.split([self.encoder_traj.pose_dim, 1], dim=-1)
            tmp = torch.zeros_like(poses_cam)
            tmp[:, :4] = F.normalize(poses_cam[:, :4])
            tmp[:, 4:] = poses_cam[:, 4:]
            poses_cam = tmp
            camPoseRel_cv2 = self.encoder_traj.toSE3(poses_cam)
            canonical_pose_cv2 = dataset.get_canonical_pose_cv2(device=device)
            canonical_extrinsics_cv2 = dataset.get_canonical_extrinsics_cv2(device=device)
            camPoses_cv2 = canonical_pose_cv2.unsqueeze(0) @ camPoseRel_cv2
            camE_cv2 = torch.inverse(camPoses_cv2).reshape(b, t - 1, 4, 4)
            camPoses_cv2 = camPoses_cv2.reshape(b, t - 1, 4, 4)
            camPoses_cv2 = torch.cat([canonical_pose_cv2.reshape(1, 1, 4, 4).repeat(b, 1, 1, 1), camPoses_cv2], dim=1)
            camE_cv2 = torch.cat([canonical_extrinsics_cv2.reshape(1, 1, 4, 4).repeat(b, 1, 1, 1), camE_cv2], dim=1)
            poses_cam_gt = sample["cam_poses_rel_cv2"][:, 1:self.N_INPUT].to(device).reshape(b * (t - 1), 4, 4)
            camPose_return = {"gt": geo_utils.mat2quat(poses_cam_gt), "pred": poses_cam, "conf": conf}
        else:
            suffix = "_canonicalized" if self.config.train.canonicalize else ""
            camE_cv2 = sample["cam_extrinsics_cv2" + suffix][:, :t].to(device)
            camPoses_cv2 = sample["cam_poses_cv2" + suffix][:, :t].to(device)
            camPose_return = None
        idxs = sequence_from_distance(camPoses_cv2[:, :, :3, 3])

        if self.config.train.parameter in ("pose", "pose_head"):                       # :98-114
            camK = sample["K_cv2"].to(device)[:, :t]
            cams = {"R": camE_cv2.reshape(b * t, 4, 4)[:, :3, :3], "T": camE_cv2.reshape(b * t, 4, 4)[:, :3, 3],
                    "K": camK.reshape(b * t, 3, 3)}
            origin_proj = self.render.proj_origin(cams, device)
            return camPose_return, 2 * origin_proj / self.config.dataset.img_size

        # cameras to render: the t input cameras + the sample's remaining (novel) GT cameras (:117-125)
        camE_all = torch.cat([camE_cv2, sample["cam_extrinsics_cv2_canonicalized"][:, self.N_INPUT:].to(device)], dim=1)
        camK = sample["K_cv2"].to(device)
        V = camE_all.shape[1]
        assert V == t_all, "sample must carry intrinsics for every rendered camera"
        cameras = {"R": camE_all.reshape(b * V, 4, 4)[:, :3, :3], "T": camE_all.reshape(b * V, 4, 4)[:, :3, 3],
                   "K": camK.reshape(b * V, 3, 3)}

        features_transformed = self.rotate(voxels=features_raw, camPoses_cv2=camPoses_cv2[:, :t], grid_size=D)
        features_transformed = chose_selected(features_transformed, idxs)

        features_mv = self.encoder_3d.fuse(features_transformed)
        densities_mv = self.encoder_3d.get_density3D(features_mv)
        features_mv = self.encoder_3d.get_render_features(features_mv)
        if self.config.dataset.name == "omniobject3d":
            densities_mv = densities_mv.clamp(min=0.0, max=1.0)

        view2vol = torch.arange(b, device=device, dtype=torch.int32)[:, None].expand(b, V).reshape(b * V).contiguous()
        rendered_imgs, rendered_masks, origin_proj = self.render(cameras, features_mv, densities_mv,
                                                                 return_origin_proj=True, view2vol=view2vol)
        if self.config.train.use_gt_pose:
            return rendered_imgs, rendered_masks
        return rendered_imgs, rendered_masks, 2 * origin_proj / self.config.dataset.img_size, camPose_return
