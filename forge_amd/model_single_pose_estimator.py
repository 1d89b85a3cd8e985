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

from . import geo_utils
from .encoder import Encoder3D
from .staging import stage_sample
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

    def reconstruct(self, features_raw, camPoses_cv2, cameras):
        """a2..a7 of models/model_single_pose_estimator.py:101-133 on given per-view feature volumes features_raw [b,t,C,D,D,D]: pose warp,
        the three fusions (first 3 views, last 2 views, all views), both heads on the three fused volumes, ray-march of the 2t cameras of
        `cameras` (scene-major: 3 cameras on the 2-view volume, 2 on the 3-view volume, t on the all-view volume) and conv_rgb.
        D = 32 is what the encoder produces; D = 64 is the reference's large grid (models/rotate.py:115-117 -> 128^3 render volume,
        BASELINE configs[3]), fed with synthetic feature volumes by tools/train_step_probe.py TRAIN_GRID=64.
        Returns (rgb [b*2t,3,img,img], masks [b*2t,1,img,img], origin_proj [b*2t,2])."""
        b, t, C, D = features_raw.shape[:4]
        device = features_raw.device
        features_transformed = self.rotate(voxels=features_raw, camPoses_cv2=camPoses_cv2, grid_size=D)

        # three fusions: first 3 views, last 2 views, all views (:108-109, :120)
        features_3v, features_2v, features_mv = self.encoder_3d.fuse_groups(
            features_transformed, [list(range(min(3, t))), list(range(max(t - 2, 0), t)), list(range(t))])
        if self.encoder_3d.training:
            # BatchNorm batch statistics (and the running-stat updates) follow the reference's call structure: the heads run on
            # cat([3v, 2v]) (:110-111) and on the all-view volume (:121-122) SEPARATELY - one 3b batch would normalise differently
            f32 = torch.cat([features_3v, features_2v], dim=0)
            r32, d32 = self.encoder_3d.heads(f32)
            rm, dm = self.encoder_3d.heads(features_mv)
            densities, features = torch.cat([d32, dm], dim=0), torch.cat([r32, rm], dim=0)
        else:
            fused = torch.cat([features_3v, features_2v, features_mv], dim=0)          # eval BN: one [3b,128,D,H,W] batch is the same arithmetic
            features, densities = self.encoder_3d.heads(fused)                         # [3b,16,2D,..], [3b,1,2D,..]
        if self.config.dataset.name == "omniobject3d":
            densities = densities.clamp(min=0.0, max=1.0)

        # view order per scene (:112-129): 2v volume x3 cams, 3v volume x2 cams, mv volume x t cams
        scene = torch.arange(b, device=device, dtype=torch.int32)[:, None]
        per_scene = torch.cat([(b + scene).expand(b, 3), scene.expand(b, 2), (2 * b + scene).expand(b, t)], dim=1)
        view2vol = per_scene.reshape(b * 2 * t).contiguous()
        return self.render(cameras, features, densities, return_origin_proj=True, view2vol=view2vol)

    def forward(self, sample, dataset, device, features_recon=None):
        """models/model_single_pose_estimator.py:26-138. `features_recon` (not in the reference): per-view feature volumes [b,t,C,D,D,D] that
        replace the encoder's in the reconstruction - BASELINE configs[3]'s 128^3-voxel scenes need D = 64 volumes, which the encoder cannot
        produce from 256^2 images (models/encoder.py:49). With GT poses the encoder is then not run at all (nothing consumes its output)."""
        sample = stage_sample(sample, device)                         # ONE pinned host->device copy for host-resident samples (f4)
        if features_recon is None:
            features_recon = sample.get("features_recon")             # ... or handed over with the sample (what a wrapped model - DDP - can be given)
        clips = sample["images"]
        b, t, c, h, w = clips.shape
        if features_recon is not None and self.config.train.use_gt_pose:
            features_raw = features_recon
        else:
            features_raw = self.encoder_3d.get_feat3D(clips.reshape(b * t, c, h, w))      # [b*t,C,D,H,W]
            _, C, D, H, W = features_raw.shape
            features_raw = features_raw.reshape(b, t, C, D, H, W)

        if not self.config.train.use_gt_pose:
            pose_vec, conf = self.encoder_traj(features_raw)                           # :45
            pose_vec, camPoses_cv2, camE_cv2 = geo_utils.predicted_camera_chain(
                pose_vec, self.encoder_traj.toSE3, *geo_utils.canonical_cameras(self, dataset, device), b, t)
            gt_rel = sample["cam_poses_rel_cv2"][:, 1:].reshape(b * (t - 1), 4, 4)
            camPose_return = {"gt": geo_utils.mat2quat(gt_rel), "pred": pose_vec, "conf": conf}
        else:
            suffix = "_canonicalized" if self.config.train.canonicalize else ""
            camE_cv2 = sample["cam_extrinsics_cv2" + suffix]
            camPoses_cv2 = sample["cam_poses_cv2" + suffix]
            camPose_return = None

        # cameras for rendering: every input camera twice (:77-85)
        cameras = geo_utils.camera_dict(camE_cv2.repeat(1, 2, 1, 1), sample["K_cv2"].repeat(1, 2, 1, 1))

        if self.config.train.parameter == "pose":                                      # :87-99
            origin_proj = self.render.proj_origin(cameras, device)
            return camPose_return, 2 * origin_proj / self.config.dataset.img_size

        rendered_imgs, rendered_masks, origin_proj = self.reconstruct(features_raw if features_recon is None else features_recon, camPoses_cv2[:, :t], cameras)
        if self.config.train.use_gt_pose:
            return rendered_imgs, rendered_masks
        return rendered_imgs, rendered_masks, 2 * origin_proj / self.config.dataset.img_size, camPose_return
