"""Pose parameterisations <-> SE(3), host-side torch math (tiny; not on the accelerated path).

Same function names / conventions as the reference's utils/geo_utils.py (:6-137 conversions,
:140-213 mat2quat): quaternions are (w,x,y,z); every *2mat takes [B, rot_dim+3] with the
translation in the trailing 3 entries and returns [B,4,4]."""
import torch
import torch.nn.functional as F


def _se3(rot, trans):
    B = rot.shape[0]
    top = torch.cat([rot, trans.reshape(B, 3, 1)], dim=2)
    bottom = torch.zeros(B, 1, 4, dtype=rot.dtype, device=rot.device)        # built on the device (no host->device copy: capturable)
    bottom[..., 3] = 1.0
    return torch.cat([top, bottom], dim=1)


def euler2mat(angle):
    """utils/geo_utils.py:6-47: R = Rz(angle[:,2]) @ Ry(angle[:,0]) @ Rx(angle[:,1]); t = angle[:,3:]"""
    x, y, z = angle[:, 1], angle[:, 0], angle[:, 2]
    zero, one = torch.zeros_like(z), torch.ones_like(z)
    cz, sz, cy, sy, cx, sx = z.cos(), z.sin(), y.cos(), y.sin(), x.cos(), x.sin()
    Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).reshape(-1, 3, 3)
    Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).reshape(-1, 3, 3)
    Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).reshape(-1, 3, 3)
    return _se3(Rz @ Ry @ Rx, angle[:, 3:])


def symmetric_orthogonalization(x):
    """utils/geo_utils.py:73-85: nearest rotation (SVD, det-corrected) of a 3x3 given as 9 numbers."""
    m = x.reshape(-1, 3, 3)
    u, _, v = torch.svd(m)
    vt = v.transpose(1, 2)
    det = torch.det(u @ vt).reshape(-1, 1, 1)
    vt = torch.cat([vt[:, :2, :], vt[:, 2:, :] * det], dim=1)
    return u @ vt


def rot9d2mat(x):
    return _se3(symmetric_orthogonalization(x[:, :9]), x[:, 9:])


def rot6d2mat(x):
    """Zhou et al. CVPR'19 (utils/geo_utils.py:89-107): Gram-Schmidt on two 3-vectors, columns b1 b2 b3."""
    b1 = F.normalize(x[:, 0:3])
    a2 = x[:, 3:6]
    b2 = F.normalize(a2 - (b1 * a2).sum(dim=1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return _se3(torch.stack([b1, b2, b3], dim=-1), x[:, 6:])


def quat2mat_transform(quat):
    """utils/geo_utils.py:122-137"""
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q.unbind(dim=1)
    rows = [w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
            2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
            2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def quat2mat(x):
    return _se3(quat2mat_transform(x[:, :4]), x[:, 4:])


def mat2quat_transform(R, eps=1e-6):
    """utils/geo_utils.py:148-213 (the torchgeometry 4-branch algorithm), as a torch.where select.
    R [B,3,3] -> (w,x,y,z)."""
    r00, r11, r22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    s = lambda i, j: R[:, i, j]
    t0 = 1 + r00 - r11 - r22
    t1 = 1 - r00 + r11 - r22
    t2 = 1 - r00 - r11 + r22
    t3 = 1 + r00 + r11 + r22
    q0 = torch.stack([s(2, 1) - s(1, 2), t0, s(1, 0) + s(0, 1), s(0, 2) + s(2, 0)], dim=-1)
    q1 = torch.stack([s(0, 2) - s(2, 0), s(1, 0) + s(0, 1), t1, s(2, 1) + s(1, 2)], dim=-1)
    q2 = torch.stack([s(1, 0) - s(0, 1), s(0, 2) + s(2, 0), s(2, 1) + s(1, 2), t2], dim=-1)
    q3 = torch.stack([t3, s(2, 1) - s(1, 2), s(0, 2) - s(2, 0), s(1, 0) - s(0, 1)], dim=-1)
    small_r22 = (r22 < eps)[:, None]
    pick_a = (r00 > r11)[:, None]
    pick_b = (r00 < -r11)[:, None]
    q = torch.where(small_r22, torch.where(pick_a, q0, q1), torch.where(pick_b, q2, q3))
    t = torch.where(small_r22, torch.where(pick_a, t0[:, None], t1[:, None]),
                    torch.where(pick_b, t2[:, None], t3[:, None]))
    return 0.5 * q / torch.sqrt(t)


def mat2quat(x):
    """[B,4,4] -> [B,7] (quaternion, translation) — utils/geo_utils.py:140-145"""
    return torch.cat([mat2quat_transform(x[:, :3, :3]), x[:, :3, 3]], dim=1)


def get_relative_pose(cam_1, cam_2):
    """utils/geo_utils.py:232-265: T_c1->c2 = inv(P1) @ P2 for rigid poses. cam_1 [4,4] or [t,4,4], cam_2 [t,4,4]"""
    if cam_1.dim() == 2:
        cam_1 = cam_1[None].expand(cam_2.shape[0], 4, 4)
    R1t = cam_1[:, :3, :3].transpose(1, 2)
    R = R1t @ cam_2[:, :3, :3]
    t = (R1t @ (cam_2[:, :3, 3] - cam_1[:, :3, 3])[..., None])[..., 0]
    return _se3(R, t)


def canonicalize_poses(canonical_pose, cam_poses_rel):
    """utils/geo_utils.py:268-287"""
    return canonical_pose[None] @ cam_poses_rel


def inverse_affine(P):
    """Inverse of affine 4x4 matrices [..., 4, 4] whose last row is (0, 0, 0, 1) - camera poses / extrinsics - in closed form:
    A^-1 = [b x c, c x a, a x b] / det for the rows a, b, c of the 3x3 block, t' = -A^-1 t. Differentiable torch ops without
    the LU + info check (a host synchronisation per call) of torch.inverse, so a loop that inverts poses every iteration
    (pose refinement, kubric_eval.py:412-530) stays asynchronous and can be captured into a hipGraph."""
    A, t = P[..., :3, :3], P[..., :3, 3]
    a, b, c = A[..., 0, :], A[..., 1, :], A[..., 2, :]
    bc, ca, ab = torch.cross(b, c, dim=-1), torch.cross(c, a, dim=-1), torch.cross(a, b, dim=-1)
    det = (a * bc).sum(dim=-1)[..., None, None]
    Ai = torch.stack([bc, ca, ab], dim=-1) / det
    ti = -(Ai @ t[..., None])
    top = torch.cat([Ai, ti], dim=-1)
    bottom = torch.zeros_like(top[..., :1, :])
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], dim=-2)


def predicted_camera_chain(pose_vec, to_se3, canonical_pose, canonical_extrinsics, b, t):
    """Cameras of a scene from the pose heads' output (models/model.py:66-81, models/model_single_pose_estimator.py:45-60).

    pose_vec [b(t-1), pose_dim]: relative pose of views 1..t-1 w.r.t. view 0; its first four entries are L2-normalised before the
    conversion whatever the rotation representation (the reference does exactly that). View 0 is the canonical camera.
    Returns (normalised pose_vec, camera->world poses [b,t,4,4], world->camera extrinsics [b,t,4,4]); the extrinsics come from the
    closed-form affine inverse, so the chain carries gradients, needs no host synchronisation and can be captured into a hipGraph."""
    pose_vec = torch.cat([F.normalize(pose_vec[:, :4]), pose_vec[:, 4:]], dim=1)
    world_from_cam = (canonical_pose[None] @ to_se3(pose_vec)).reshape(b, t - 1, 4, 4)
    first = lambda m: m.reshape(1, 1, 4, 4).expand(b, 1, 4, 4)
    poses = torch.cat([first(canonical_pose), world_from_cam], dim=1)
    extrinsics = torch.cat([first(canonical_extrinsics), inverse_affine(world_from_cam)], dim=1)
    return pose_vec, poses, extrinsics


def canonical_cameras(owner, dataset, device):
    """(canonical pose, canonical extrinsics) of `dataset` on `device` (models/model.py:74-75), fetched ONCE per (dataset, device) and kept on
    the model `owner`: the dataset hands out host tensors, and a pageable host->device copy per forward synchronises the stream and cannot be
    captured into a hipGraph. The cached copies are made with inference mode OFF: a first forward under torch.inference_mode() would otherwise
    cache inference tensors, which a later training forward cannot save for backward."""
    cache = owner.__dict__.setdefault("_canon", {})
    key = (id(dataset), str(device))
    if key not in cache:
        cache.clear()                                                   # one dataset at a time; the entry keeps the dataset alive, so its id stays unique
        with torch.inference_mode(False):
            own = lambda t: t.to(device=device, dtype=torch.float32).clone()      # clone: a normal tensor even if the dataset handed out an inference tensor
            cache[key] = (own(dataset.get_canonical_pose_cv2(device=device)), own(dataset.get_canonical_extrinsics_cv2(device=device)), dataset)
    return cache[key][:2]


def camera_dict(extrinsics, K):
    """The {'R','T','K'} dict VolRender.forward takes (models/volume_render.py:40-48) from [..,4,4] extrinsics and [..,3,3] intrinsics."""
    E = extrinsics.reshape(-1, 4, 4)
    return {"R": E[:, :3, :3], "T": E[:, :3, 3], "K": K.reshape(-1, 3, 3)}
