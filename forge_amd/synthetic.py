"""Synthetic scenes in the Kubric sample-dict schema + seeded weights.

There is no dataset or checkpoint in the build/bench environment, so benchmarks, smoke and
parity tests run on synthetic data of the reference's shapes:

* sample dict keys/shapes follow dataset/kubric.py:390-402 (`images`, `fg_probabilities`,
  `K_cv2`, `cam_extrinsics_cv2_canonicalized`, `cam_poses_cv2_canonicalized`, `cam_poses_rel_cv2`);
* camera conventions follow dataset/kubric.py:78-104 (OpenCV frame; canonical camera = identity
  rotation, t_z = render.camera_z) and utils/geo_utils.py:232-287 (relative / canonicalised poses);
* intrinsics are demo.py:39-41: K = [[1.38888,0,0.5],[0,1.38888,0.5],[0,0,1]] * img_size (row 2 kept 1).

Weights are generated per state-dict key from a numpy MT19937 stream seeded by crc32(key)^seed,
so the golden-vector generator (oracle/make_golden.py), the tests and the bench all see the same
numbers without committing 220 MB of parameters.
"""
import math
import zlib

import numpy as np
import torch


class SimpleConfig(dict):
    """Attribute-style nested config (the reference uses EasyDict, config/config.py:6-79).
    Any object exposing the same attribute paths works with forge_amd modules."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = SimpleConfig(v) if isinstance(v, dict) and not isinstance(v, SimpleConfig) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def kubric_config(img_size=256, volume_size=1.0, n_pts_per_ray=64, min_depth=0.5, max_depth=2.0,
                  camera_z=1.5, use_gt_pose=True, parameter="all", dataset_name="kubric"):
    """config/kubric/gt_pose.yaml:10-60 (only the keys the model reads, SURVEY.md §5 "Config")."""
    return SimpleConfig({
        "dataset": {"name": dataset_name, "img_size": img_size, "num_frame": 5},
        "network": {"padding_mode": "zeros", "rot_representation": "quat",
                    "scale_rotate": 0.01, "scale_translate": 0.01, "backbone": "resnet"},
        "render": {"n_pts_per_ray": n_pts_per_ray, "volume_size": volume_size, "min_depth": min_depth,
                   "max_depth": max_depth, "camera_z": camera_z, "k_size": 5},
        "train": {"use_gt_pose": use_gt_pose, "canonicalize": True, "parameter": parameter, "lr": 0.0008, "accumulation_step": 1,
                  "adjust_iter_num": [17500, 26000, 34000, 50000]},
        "loss": {"recon_rgb": 5.0, "recon_mask": 1.0, "perceptual_img": 0.0, "regu_origin_proj": 0.0},
    })


class SyntheticDataset:
    """Stand-in for the `dataset` argument of forward(): only the two methods the model calls
    (models/model.py:74-75; dataset/kubric.py:100-104, 448-452)."""

    def __init__(self, camera_z=1.5):
        self.canonical_extrinsics_cv2 = torch.eye(4)
        self.canonical_extrinsics_cv2[2, 3] = camera_z
        self.canonical_pose_cv2 = torch.inverse(self.canonical_extrinsics_cv2)

    def get_canonical_extrinsics_cv2(self, device="cpu"):
        return self.canonical_extrinsics_cv2.to(device)

    def get_canonical_pose_cv2(self, device="cpu"):
        return self.canonical_pose_cv2.to(device)


def intrinsics(img_size=256):
    """demo.py:39-41"""
    f = 1.38888 * img_size
    return torch.tensor([[f, 0.0, 0.5 * img_size], [0.0, f, 0.5 * img_size], [0.0, 0.0, 1.0]])


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, 0.0, s, 0.0], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]])


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, c, -s, 0.0], [0.0, s, c, 0.0], [0.0, 0.0, 0.0, 1.0]])


def orbit_cameras(n_views=10, camera_z=1.5, elev_deg=0.0, jitter=None):
    """Canonical view 0 + orbit about the object centre: first 5 views at azimuth
    72*i deg, next 5 ("novel") at 36+72*i deg. Relative pose of view i in view 0's frame is
    T_c R T_c^-1 with T_c the translation to the object centre (0,0,camera_z); canonicalised pose
    = P_can @ rel (utils/geo_utils.py:268-287). Returns (poses[n,4,4], extrinsics[n,4,4], rel[n,4,4])."""
    can_e = torch.eye(4)
    can_e[2, 3] = camera_z
    can_p = torch.inverse(can_e)
    Tc = torch.eye(4)
    Tc[2, 3] = camera_z
    Tci = torch.inverse(Tc)
    rels = []
    for i in range(n_views):
        az = math.radians(72.0 * i if i < 5 else 36.0 + 72.0 * (i - 5))
        el = math.radians(elev_deg * ((i % 3) - 1))
        R = _rot_y(az) @ _rot_x(el)
        if jitter is not None and i > 0:
            R = R @ _rot_y(float(jitter[i, 0])) @ _rot_x(float(jitter[i, 1]))
        rels.append(Tc @ R @ Tci)
    rel = torch.stack(rels)
    rel[0] = torch.eye(4)
    poses = can_p[None] @ rel
    return poses, torch.inverse(poses), rel


def make_sample(b=1, n_views=10, img_size=256, camera_z=1.5, seed=0, elev_deg=10.0):
    """A `sample` dict with the dataset/kubric.py:390-402 schema. images = U[0,1)*mask with mask a
    projected ellipsoid at the object centre."""
    g = torch.Generator().manual_seed(seed)
    K = intrinsics(img_size)
    out = {k: [] for k in ("images", "fg_probabilities", "K_cv2", "cam_extrinsics_cv2_canonicalized",
                           "cam_poses_cv2_canonicalized", "cam_poses_rel_cv2")}
    ys, xs = torch.meshgrid(torch.arange(img_size, dtype=torch.float32) + 0.5,
                            torch.arange(img_size, dtype=torch.float32) + 0.5, indexing="ij")
    for s in range(b):
        jit = (torch.rand(n_views, 2, generator=g) - 0.5) * 0.2
        poses, extr, rel = orbit_cameras(n_views, camera_z, elev_deg, jit)
        radii = 0.25 + 0.15 * torch.rand(3, generator=g)
        imgs, masks = [], []
        for v in range(n_views):
            # silhouette of the ellipsoid x^T A x = 1 seen from camera v: rays o + d*l hit iff disc >= 0
            Rcw = poses[v, :3, :3]
            o = poses[v, :3, 3]
            d_cam = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], dim=-1)
            d = d_cam @ Rcw.T
            A = 1.0 / radii ** 2
            a = (d * d * A).sum(-1)
            bq = (d * o * A).sum(-1)
            c = (o * o * A).sum() - 1.0
            mask = ((bq * bq - a * c) >= 0).float()[None]
            imgs.append(torch.rand(3, img_size, img_size, generator=g) * mask)
            masks.append(mask)
        out["images"].append(torch.stack(imgs))
        out["fg_probabilities"].append(torch.stack(masks))
        out["K_cv2"].append(K[None].repeat(n_views, 1, 1))
        out["cam_extrinsics_cv2_canonicalized"].append(extr)
        out["cam_poses_cv2_canonicalized"].append(poses)
        out["cam_poses_rel_cv2"].append(rel)
    return {k: torch.stack(v).contiguous() for k, v in out.items()}


def blob_volumes(n, D, C=16, seed=0, peak=1.5, sigma2=0.08, vol_size=1.0):
    """Renderer micro-benchmark volumes: density = peak*exp(-|x|^2/sigma2) over the
    voxel-centre world grid (peaks > 1 exercise the unclamped-density path),
    features N(0,1). Returns feat [n,C,D,D,D], dens [n,1,D,D,D]."""
    g = torch.Generator().manual_seed(seed)
    e = 0.5 * (D - 1) * (vol_size / D)
    lin = torch.linspace(-1.0, 1.0, D) * e
    Z, Y, X = torch.meshgrid(lin, lin, lin, indexing="ij")
    dens = []
    for i in range(n):
        ctr = (torch.rand(3, generator=g) - 0.5) * 0.2
        r2 = (X - ctr[0]) ** 2 + (Y - ctr[1]) ** 2 + (Z - ctr[2]) ** 2
        dens.append(peak * torch.exp(-r2 / sigma2))
    dens = torch.stack(dens)[:, None].contiguous()
    feat = torch.randn(n, C, D, D, D, generator=g)
    return feat, dens


# ------------------------------------------------------------------------------------------
# seeded weights
# ------------------------------------------------------------------------------------------
_CONVT_KEYS = ("features_head.0.weight", "density_head.0.weight", "conv_rgb.0.weight")


def seeded_state_dict(template, seed=0):
    """Deterministic values for every entry of `template` (a state_dict or {key: shape}).

    conv / linear weights ~ N(0, sqrt(2/fan_in)); BN weight U(0.5,1.5) (U(0.1,0.3) for the residual
    branch's last BN so the ResNet trunk stays O(1)); BN bias / running_mean N(0,0.1);
    running_var U(0.5,1.5) -> eval-mode BN is non-trivial; other biases N(0,0.05);
    the density head's last bias is -0.6: ~10 % of voxels get a positive ReLU density (mean ~0.05,
    peaks > 1), i.e. a sparse object rather than an all-opaque or all-empty volume."""
    keys = list(template.keys())
    keyset = set(keys)
    out = {}
    for k in keys:
        v = template[k]
        shape = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        rng = np.random.RandomState((zlib.crc32(k.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)
        stem = k.rsplit(".", 1)[0]
        is_bn = (stem + ".running_mean") in keyset
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=torch.int64)
            continue
        if k.endswith("running_mean"):
            a = rng.standard_normal(shape) * 0.1
        elif k.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif is_bn and k.endswith(".weight"):
            a = rng.uniform(0.1, 0.3, shape) if stem.endswith("bn3") else rng.uniform(0.5, 1.5, shape)
        elif is_bn and k.endswith(".bias"):
            a = rng.standard_normal(shape) * 0.1
        elif len(shape) >= 2:
            if k.endswith(_CONVT_KEYS):
                fan_in = shape[0] * int(np.prod(shape[2:])) / (2 ** (len(shape) - 2))
            else:
                fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) * math.sqrt(2.0 / max(fan_in, 1))
        elif k.endswith("density_head.6.bias"):
            a = np.full(shape, -0.6)
        elif k.endswith(".weight"):          # LayerNorm-style 1-D scale
            a = rng.uniform(0.5, 1.5, shape)
        else:
            a = rng.standard_normal(shape) * 0.05
        out[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out
