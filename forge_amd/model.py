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

from . import geo_utils
from .encoder import Encoder3D
from .staging import stage_sample
from .pose_estimator_2d import PoseEstimator2D
from .pose_estimator_3d import PoseEstimator3D
from .rotate import Rotate_world
from .volume_render import VolRender


def sequence_from_distance(trans):
    """models/model.py:152-158 — translations [b,t,3] -> view order by squared distance to view 0."""
    dist = ((trans - trans[:, 0:1, :]) ** 2).sum(dim=-1)
    return torch.sort(dist, descending=False, stable=True)[1]      # stable: equal distances keep their view order (deterministic)


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

    def reconstruct(self, features_raw, camPoses_cv2, cameras, idxs=None):
        """a2..a7 on given per-view feature volumes: pose warp -> view order -> ConvGRU fusion -> heads -> ray-march of the V cameras of
        `cameras` (b*V entries, scene-major) -> conv_rgb. features_raw [b,t,C,D,D,D] with D in Rotate_world's grid sizes: 32 is what
        the encoder produces from 256^2 images (models/encoder.py:49); 64 is the reference's large-grid path (models/rotate.py:115-117)
        -> 128^3 render volume (BASELINE configs[3]/[4]), fed with synthetic feature volumes by bench.py --grid 64.
        Returns (rgb [b*V,3,img,img], masks [b*V,1,img,img], origin_proj [b*V,2])."""
        b, t, C, D = features_raw.shape[:4]
        device = features_raw.device
        # warp + view ordering (models/model.py:127-128) in one launch; idxs = None: by distance to view 0, ranked on the device
        features_transformed = self.rotate(voxels=features_raw, camPoses_cv2=camPoses_cv2, grid_size=D, order="distance" if idxs is None else idxs)
        features_mv, densities_mv = self.encoder_3d.heads(self.encoder_3d.fuse(features_transformed))
        if self.config.dataset.name == "omniobject3d":
            densities_mv = densities_mv.clamp(min=0.0, max=1.0)
        V = cameras["R"].shape[0] // b
        return self.render(cameras, features_mv, densities_mv, return_origin_proj=True, view2vol=self._view2vol(b, V, device))

    def _view2vol(self, b, V, device):
        """scene index of every rendered view (b*V int32), built once per (b, V, device): the fused volume of a scene is rendered by its V
        cameras through this index instead of being repeated V times (models/model.py:138-139)."""
        cache = self.__dict__.setdefault("_v2v", {})
        key = (b, V, str(device))
        if key not in cache:
            cache[key] = torch.arange(b, device=device, dtype=torch.int32)[:, None].expand(b, V).reshape(b * V).contiguous()
        return cache[key]

    # The 2-D pose estimator needs only the images: on the MI355X it is launched on a side HIP stream BEFORE the encoder, so that its ~60 small ResNet
    # launches (5 images: 320-1280 workgroups each) share the chip with the encoder trunk's equally under-filled ones - forward, and backward too
    # (autograd replays each node on its forward stream). False = one stream.
    pose2d_side_stream = True

    def _pose2d_features(self, clips):
        """encoder_traj_2d(clips, return_features=True), launched on the side stream when enabled; returns (features, join) - call join() on the
        consumer's stream before the features are used."""
        if not (self.pose2d_side_stream and clips.is_cuda):
            return self.encoder_traj_2d(clips, return_features=True), (lambda: None)
        cur = torch.cuda.current_stream(clips.device)
        side = self.__dict__.setdefault("_side_streams", {}).setdefault(str(clips.device), None) or torch.cuda.Stream(device=clips.device)
        self.__dict__["_side_streams"][str(clips.device)] = side
        side.wait_stream(cur)                                           # the images (and the parameters' last update) are complete
        with torch.cuda.stream(side):
            feat = self.encoder_traj_2d(clips, return_features=True)

        def join():
            cur.wait_stream(side)
            feat.record_stream(cur)                                     # allocated on the side stream, consumed (and later freed) on this one
        return feat, join

    def predict_poses(self, features_raw, clips, sample, dataset, device, pose_feat_2d=None, join_2d=None):
        """models/model.py:60-84 - relative poses of views 1..t-1 from the 3-D pose estimator (on the per-view feature volumes) and the 2-D pose
        estimator (on the images), joined by the pose head; quaternion normalised, toSE3, chained onto the canonical camera.
        pose_feat_2d / join_2d: the 2-D estimator's features when the caller launched it on a side stream (_pose2d_features) and the join to call
        before they are consumed - after the 3-D estimator's launches, so that the side stream's tail runs beside them.
        Returns (camPoses_cv2 [b,t,4,4], camE_cv2 [b,t,4,4], {'gt', 'pred', 'conf'})."""
        b, t = features_raw.shape[:2]
        if pose_feat_2d is None:
            pose_feat_2d = self.encoder_traj_2d(clips, return_features=True)
        pose_feat_3d = self.encoder_traj(features_raw, return_features=True)
        if join_2d is not None:
            join_2d()
        pose_feat = torch.cat([pose_feat_3d, pose_feat_2d], dim=-1)                                               # [b(t-1),1024] each
        pose_vec, conf = self.pose_head(pose_feat).split([self.encoder_traj.pose_dim, 1], dim=-1)
        pose_vec, camPoses_cv2, camE_cv2 = geo_utils.predicted_camera_chain(pose_vec, self.encoder_traj.toSE3, *geo_utils.canonical_cameras(self, dataset, device), b, t)
        gt_rel = sample["cam_poses_rel_cv2"][:, 1:self.N_INPUT].reshape(b * (t - 1), 4, 4)
        return camPoses_cv2, camE_cv2, {"gt": geo_utils.mat2quat(gt_rel), "pred": pose_vec, "conf": conf}

    def forward(self, sample, dataset, device, features_recon=None):
        """models/model.py:42-148. `features_recon` (not in the reference): per-view feature volumes [b,5,C,D,D,D] that replace the encoder's
        in the reconstruction (rotate -> fuse -> heads -> render) while the pose estimators keep their native inputs - BASELINE configs[4]'s
        128^3-voxel scenes need D = 64 volumes, which the encoder cannot produce from 256^2 images (models/encoder.py:49)."""
        sample = stage_sample(sample, device)                         # ONE pinned host->device copy for host-resident samples (f4)
        if features_recon is None:
            features_recon = sample.get("features_recon")             # ... or handed over with the sample (what a wrapped model - DDP - can be given)
        b, t_all = sample["images"].shape[:2]
        clips = sample["images"][:, :self.N_INPUT]
        b, t, c, h, w = clips.shape
        f2d, join2d = (None, None) if self.config.train.use_gt_pose else self._pose2d_features(clips)
        features_raw = self.encoder_3d.get_feat3D(clips.reshape(b * t, c, h, w))
        _, C, D, H, W = features_raw.shape
        features_raw = features_raw.reshape(b, t, C, D, H, W)

        if not self.config.train.use_gt_pose:
            camPoses_cv2, camE_cv2, camPose_return = self.predict_poses(features_raw, clips, sample, dataset, device, pose_feat_2d=f2d, join_2d=join2d)
        else:
            suffix = "_canonicalized" if self.config.train.canonicalize else ""
            camE_cv2 = sample["cam_extrinsics_cv2" + suffix][:, :t]
            camPoses_cv2 = sample["cam_poses_cv2" + suffix][:, :t]
            camPose_return = None

        if self.config.train.parameter in ("pose", "pose_head"):                       # :98-114
            origin_proj = self.render.proj_origin(geo_utils.camera_dict(camE_cv2, sample["K_cv2"][:, :t]), device)
            return camPose_return, 2 * origin_proj / self.config.dataset.img_size

        # cameras to render: the t input cameras + the sample's remaining (novel) GT cameras (:117-125)
        camE_all = torch.cat([camE_cv2, sample["cam_extrinsics_cv2_canonicalized"][:, self.N_INPUT:]], dim=1)
        V = camE_all.shape[1]
        assert V == t_all, "sample must carry intrinsics for every rendered camera"
        cameras = geo_utils.camera_dict(camE_all, sample["K_cv2"])

        rendered_imgs, rendered_masks, origin_proj = self.reconstruct(features_raw if features_recon is None else features_recon,
                                                                      camPoses_cv2[:, :t], cameras)                      # views ordered by distance
        if self.config.train.use_gt_pose:
            return rendered_imgs, rendered_masks
        return rendered_imgs, rendered_masks, 2 * origin_proj / self.config.dataset.img_size, camPose_return
