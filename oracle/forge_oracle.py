"""TEST INFRASTRUCTURE — CPU oracle for the FORGE reconstruction hot path.

A functional restatement, in plain torch-CPU fp32 ops, of the reference algorithm
(/root/reference, pure Python on PyTorch + PyTorch3D 0.7.0). Every function cites the
reference file:line it follows. It is the CHECKER for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg import it. Nothing under
`forge_amd/` imports, calls or falls back to it.

Pinning status (see DESIGN.md "Oracle"):
  * rotate / ConvGRU fusion / heads / conv_rgb / orchestration: PINNED against the reference's
    own module code, imported in the build container through `oracle/ref_import.py`
    (golden vectors in tests/golden/, generator `oracle/make_golden.py`).
  * PyTorch3D arithmetic (ray sampler, volume sampler, EA ray-marcher, camera conversion):
    PyTorch3D itself is not installable offline, so those are pinned only against
    (i) `oracle/shims/pytorch3d`, an independently-structured restatement of the published
    0.7.0 algorithms (4x4 transforms + matrix inverse + cumprod), and (ii) the known answers
    in the reference text (grid half-extent 0.4844 models/rotate.py:23; origin projects to the
    image centre scripts/kubric_compute_loss.py:60-62), and (iii) since round 2 the analytic
    known answers of tests/kat_render.py, derived in float64 from the documented contracts of
    those PyTorch3D pieces with neither this file nor the shims (uniform slabs on cubic and
    anisotropic grids, single-voxel impulses under a rotated camera with fx != fy, cx != cy).
    => still "parity unpinned" w.r.t. the real PyTorch3D BINARY (never run here), pinned w.r.t.
    its documented semantics.
  * predicted-pose orchestration (models/model.py:58-96 and the pose estimators it calls) is
    not restated here: forge_amd's stock-torch pose estimators are compared directly with the
    reference's outputs (tests/golden/forward_joint.npz).

Weights are passed as a flat dict with the reference's state_dict key names
(SURVEY.md Appendix B), e.g. "encoder_3d.fusion_feature.cells.0.conv_gate.weight".
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm default
LRELU = 0.01           # nn.LeakyReLU default


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def _bn(x, w, prefix, training=False):
    """nn.BatchNorm{2,3}d: eval mode uses running stats; train mode uses biased batch stats."""
    if training:
        return F.batch_norm(x, None, None, w[prefix + ".weight"], w[prefix + ".bias"], True, 0.1, BN_EPS)
    return F.batch_norm(x, w[prefix + ".running_mean"], w[prefix + ".running_var"],
                        w[prefix + ".weight"], w[prefix + ".bias"], False, 0.1, BN_EPS)


def grid_half_extent(D, vol_size=1.0):
    """pytorch3d Volumes.get_coord_grid(world_coordinates=True).max(): 0.5*(D-1)*(vol/D).
    models/rotate.py:22-23 ("should be 0.4844" for D=32, vol=1)."""
    return 0.5 * (D - 1) * (vol_size / D)


# --------------------------------------------------------------------------------------
# a2  rotate  (models/rotate.py:48-61, 64-89, 92-141)
# --------------------------------------------------------------------------------------
def relative_transforms(cam_poses):
    """models/rotate.py:78-89: T = P_0 @ inverse(P_i) for i = 1..t-1.  [B,t,4,4] -> [B*(t-1),4,4]"""
    B, t = cam_poses.shape[:2]
    p0 = cam_poses[:, 0:1].repeat(1, t - 1, 1, 1).reshape(B * (t - 1), 4, 4)
    p1 = cam_poses[:, 1:].reshape(B * (t - 1), 4, 4)
    return p0 @ torch.inverse(p1)


def rotate_world(voxels, cam_poses, vol_size=1.0):
    """models/rotate.py:92-156 (grid_size = D). voxels [B,t,C,D,H,W], cam_poses [B,t,4,4].

    Sample grid = (G_world @ T^T)[:, :3] / grid_coord_max with G_world the voxel-centre world
    coordinates (x,y,z) (rotate.py:127-135); F.grid_sample default = trilinear, zeros padding,
    align_corners=False (rotate.py:137-138) although the normalisation is an align_corners=True
    one — reproduced on purpose. View 0 passes through (rotate.py:141).
    """
    B, t, C, D, H, W = voxels.shape
    assert D == H == W, "reference grids are cubic (rotate.py:18-35)"
    e = grid_half_extent(D, vol_size)
    lin = torch.linspace(-1.0, 1.0, D, dtype=voxels.dtype)
    Z, Y, X = torch.meshgrid(lin, lin, lin, indexing="ij")
    gw = torch.stack([X, Y, Z, torch.zeros_like(X)], dim=-1) * e      # world coords, [D,H,W,4]
    gw[..., 3] = 1.0
    T = relative_transforms(cam_poses).to(voxels.dtype)               # [n,4,4]
    n = T.shape[0]
    pos = gw.reshape(1, -1, 4).repeat(n, 1, 1)
    cam = torch.matmul(pos, T.permute(0, 2, 1))[:, :, :3]
    grid = (cam / e).reshape(n, D, H, W, 3)
    warped = F.grid_sample(voxels[:, 1:].reshape(n, C, D, H, W), grid, mode="bilinear",
                           padding_mode="zeros", align_corners=False)
    warped = warped.reshape(B, t - 1, C, D, H, W)
    return torch.cat([voxels[:, 0:1], warped], dim=1)


# --------------------------------------------------------------------------------------
# a3  view ordering  (models/model.py:152-168)
# --------------------------------------------------------------------------------------
def sequence_from_distance(trans):
    """models/model.py:152-158: argsort of squared distance to view 0's translation. [b,t,3]->[b,t]"""
    dist = ((trans - trans[:, 0:1, :]) ** 2).sum(dim=-1)
    return torch.sort(dist, descending=False)[1]


def chose_selected(tensor, idxs):
    """models/model.py:161-168"""
    return torch.stack([tensor[i][idxs[i]] for i in range(len(idxs))])


# --------------------------------------------------------------------------------------
# a4  fusion  (models/fusion.py:21-35, 61-68, 71-95; models/encoder.py:59-63)
# --------------------------------------------------------------------------------------
def conv_gru_cell(x, h, w, prefix):
    """models/fusion.py:29-35. Gate split order is (update, reset)."""
    hid = h.shape[1]
    g = F.conv3d(torch.cat([x, h], dim=1), w[prefix + ".conv_gate.weight"], w[prefix + ".conv_gate.bias"], padding=1)
    update, reset = torch.split(g, hid, dim=1)
    update, reset = torch.sigmoid(update), torch.sigmoid(reset)
    cand = torch.tanh(F.conv3d(torch.cat([x, h * reset], dim=1), w[prefix + ".out_gate.weight"],
                               w[prefix + ".out_gate.bias"], padding=1))
    return h * (1 - update) + cand * update


def fuse(x, w, prefix="encoder_3d.fusion_feature", training=False):
    """Encoder3D.fuse (models/encoder.py:59-63) -> ConvGRU_3D.forward (models/fusion.py:71-95).
    x [b,t,C,D,H,W] -> [b,C,D,H,W]. h0 = fusion_conv(mean_t x); sequential GRU over t; BN."""
    m = x.mean(dim=1)
    h = F.conv3d(m, w[prefix + ".fusion_conv.0.weight"], w[prefix + ".fusion_conv.0.bias"], padding=1)
    h = F.leaky_relu(_bn(h, w, prefix + ".fusion_conv.1", training), LRELU)
    h = F.conv3d(h, w[prefix + ".fusion_conv.3.weight"], w[prefix + ".fusion_conv.3.bias"], padding=1)
    h = F.leaky_relu(_bn(h, w, prefix + ".fusion_conv.4", training), LRELU)
    for t in range(x.shape[1]):
        h = conv_gru_cell(x[:, t], h, w, prefix + ".cells.0")
    return _bn(h, w, prefix + ".fusion_norm", training)


# --------------------------------------------------------------------------------------
# a5  heads  (models/encoder.py:16-34, 53-57)
# --------------------------------------------------------------------------------------
def render_features_head(z, w, prefix="encoder_3d.features_head", training=False):
    """models/encoder.py:16-22: ConvT3d(128,32,4,s2,p1) BN LReLU Conv3d(32,16,3,p1) BN"""
    y = F.conv_transpose3d(z, w[prefix + ".0.weight"], w[prefix + ".0.bias"], stride=2, padding=1)
    y = F.leaky_relu(_bn(y, w, prefix + ".1", training), LRELU)
    y = F.conv3d(y, w[prefix + ".3.weight"], w[prefix + ".3.bias"], padding=1)
    return _bn(y, w, prefix + ".4", training)


def density_head(z, w, prefix="encoder_3d.density_head", training=False):
    """models/encoder.py:25-34: ConvT3d BN LReLU Conv3d(32,8) BN LReLU Conv3d(8,1) ReLU"""
    y = F.conv_transpose3d(z, w[prefix + ".0.weight"], w[prefix + ".0.bias"], stride=2, padding=1)
    y = F.leaky_relu(_bn(y, w, prefix + ".1", training), LRELU)
    y = F.conv3d(y, w[prefix + ".3.weight"], w[prefix + ".3.bias"], padding=1)
    y = F.leaky_relu(_bn(y, w, prefix + ".4", training), LRELU)
    y = F.conv3d(y, w[prefix + ".6.weight"], w[prefix + ".6.bias"], padding=1)
    return F.relu(y)


# --------------------------------------------------------------------------------------
# a1  encoder  (models/encoder.py:46-51, 71-78)
# --------------------------------------------------------------------------------------
_RESNET_LAYERS = [(4, 3, 64, 1), (5, 4, 128, 2), (6, 6, 256, 1), (7, 3, 512, 1)]  # (seq idx, blocks, planes, stride)


def resnet50_trunk(img, w, prefix="encoder_3d.feature_extraction", training=False):
    """torchvision ResNet-50 children[:-2] with layer3[0]/layer4[0] conv2+downsample stride -> 1
    (models/encoder.py:71-78). img [N,3,H,W] -> [N,2048,H/8,W/8]."""
    p = prefix
    x = F.conv2d(img, w[p + ".0.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, w, p + ".1", training))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for idx, blocks, planes, stride in _RESNET_LAYERS:
        for b in range(blocks):
            q = "%s.%d.%d" % (p, idx, b)
            s = stride if b == 0 else 1
            out = F.relu(_bn(F.conv2d(x, w[q + ".conv1.weight"]), w, q + ".bn1", training))
            out = F.relu(_bn(F.conv2d(out, w[q + ".conv2.weight"], stride=s, padding=1), w, q + ".bn2", training))
            out = _bn(F.conv2d(out, w[q + ".conv3.weight"]), w, q + ".bn3", training)
            if (q + ".downsample.0.weight") in w:
                idn = _bn(F.conv2d(x, w[q + ".downsample.0.weight"], stride=s), w, q + ".downsample.1", training)
            else:
                idn = x
            x = F.relu(out + idn)
    return x


def get_feat3D(img, w, training=False):
    """models/encoder.py:46-51: trunk -> view(-1,64,32,H,W) (channel c = c3d*32 + z) -> conv1."""
    z2d = resnet50_trunk(img, w, training=training)
    _, _, H, W = z2d.shape
    z3d = z2d.view(-1, 64, 32, H, W)
    z3d = F.conv3d(z3d, w["encoder_3d.conv1.0.weight"], w["encoder_3d.conv1.0.bias"], padding=1)
    return F.leaky_relu(_bn(z3d, w, "encoder_3d.conv1.1", training), LRELU)


# --------------------------------------------------------------------------------------
# a6  renderer  (models/volume_render.py:40-88 + PyTorch3D 0.7.0, closed form: SURVEY.md A.2)
# --------------------------------------------------------------------------------------
def halve_intrinsics(K):
    """models/volume_render.py:50-51 (K /= 2; K[:,2,2] = 1) — on a COPY."""
    K = K.clone() / 2.0
    K[:, -1, -1] = 1.0
    return K


def ray_points(R, T, K_half, Hr, Wr, S, zmin, zmax):
    """cameras_from_opencv_projection + NDCGridRaysampler + unproject, reduced to an OpenCV
    pinhole with half-pixel centres: c = -R^T t; d_cam = ((w+.5-cx)/fx, (h+.5-cy)/fy, 1);
    p_s = c + R^T d_cam * z_s, z_s = linspace(zmin, zmax, S) (camera-z depths).
    Returns points [B,Hr,Wr,S,3] and z [S]."""
    B = R.shape[0]
    dt = R.dtype
    ws = torch.arange(Wr, dtype=dt) + 0.5
    hs = torch.arange(Hr, dtype=dt) + 0.5
    fx, fy, cx, cy = K_half[:, 0, 0], K_half[:, 1, 1], K_half[:, 0, 2], K_half[:, 1, 2]
    dx = (ws[None, None, :] - cx[:, None, None]) / fx[:, None, None]          # [B,1,Wr]
    dy = (hs[None, :, None] - cy[:, None, None]) / fy[:, None, None]          # [B,Hr,1]
    d_cam = torch.stack([dx.expand(B, Hr, Wr), dy.expand(B, Hr, Wr), torch.ones(B, Hr, Wr, dtype=dt)], dim=-1)
    Rt = R.transpose(1, 2)
    d_world = torch.einsum("bij,bhwj->bhwi", Rt, d_cam)
    c = -torch.einsum("bij,bj->bi", Rt, T)
    z = torch.linspace(zmin, zmax, S, dtype=dt)
    pts = c[:, None, None, None, :] + d_world[:, :, :, None, :] * z[None, None, None, :, None]
    return pts, z


def render_rays(feat, dens, R, T, K_half, Hr, Wr, S, zmin, zmax, vol_size=1.0, render_depth=False):
    """VolumeRenderer.__call__ (models/volume_render.py:59-63): VolumeSampler (grid_sample
    bilinear, zeros, align_corners=True in local coords p / (0.5 (D-1) vol/D)) + EA ray-marcher
    (w_s = d_s * prod_{j<s}(1 - d_j); feat = sum w f; opacity = 1 - prod(1-d); depth = sum w z
    with the README.md:26-33 patch). feat [B,C,D,H,W], dens [B,1,D,H,W] -> [B,Hr,Wr,C+1(+1)]."""
    B, C, D, H, W = feat.shape
    pts, z = ray_points(R, T, K_half, Hr, Wr, S, zmin, zmax)
    half = torch.tensor([grid_half_extent(W, vol_size * W / D), grid_half_extent(H, vol_size * H / D),
                         grid_half_extent(D, vol_size)], dtype=feat.dtype)
    local = (pts / half).view(B, -1, 1, 1, 3)
    d = F.grid_sample(dens, local, mode="bilinear", padding_mode="zeros", align_corners=True)
    f = F.grid_sample(feat, local, mode="bilinear", padding_mode="zeros", align_corners=True)
    d = d.permute(0, 2, 3, 4, 1).reshape(B, Hr, Wr, S)
    f = f.permute(0, 2, 3, 4, 1).reshape(B, Hr, Wr, S, C)
    cp = torch.cumprod(1.0 - d, dim=-1)
    trans = torch.cat([torch.ones_like(cp[..., :1]), cp[..., :-1]], dim=-1)
    wgt = d * trans
    out = [(wgt[..., None] * f).sum(dim=-2), 1.0 - torch.prod(1.0 - d, dim=-1, keepdim=True)]
    if render_depth:
        out.append((wgt * z).sum(dim=-1, keepdim=True))
    return torch.cat(out, dim=-1)


def origin_projection(R, T, K_half):
    """cameras.transform_points_screen(0) (models/volume_render.py:77-79) == OpenCV projection of
    the world origin at half resolution: (fx*tx/tz + cx, fy*ty/tz + cy)."""
    x = K_half[:, 0, 0] * T[:, 0] / T[:, 2] + K_half[:, 0, 2]
    y = K_half[:, 1, 1] * T[:, 1] / T[:, 2] + K_half[:, 1, 2]
    return torch.stack([x, y], dim=-1)


# --------------------------------------------------------------------------------------
# a7  neural up-sampler  (models/volume_render.py:29-37, 71-74)
# --------------------------------------------------------------------------------------
def conv_rgb(x, w, prefix="render.conv_rgb", k_size=5, training=False):
    """ConvT2d(16,16,k+1,s2,p=k//2) BN LReLU Conv2d(16,8,k) BN LReLU Conv2d(8,3,k) then ReLU (:73)."""
    p = k_size // 2
    y = F.conv_transpose2d(x, w[prefix + ".0.weight"], w[prefix + ".0.bias"], stride=2, padding=p)
    y = F.leaky_relu(_bn(y, w, prefix + ".1", training), LRELU)
    y = F.conv2d(y, w[prefix + ".3.weight"], w[prefix + ".3.bias"], padding=p)
    y = F.leaky_relu(_bn(y, w, prefix + ".4", training), LRELU)
    y = F.conv2d(y, w[prefix + ".6.weight"], w[prefix + ".6.bias"], padding=p)
    return F.relu(y)


def vol_render(feat, dens, R, T, K, w, img_size, S, zmin, zmax, vol_size=1.0, k_size=5,
               render_depth=False, return_origin_proj=False, training=False):
    """VolRender.forward (models/volume_render.py:40-88), K is the FULL-resolution intrinsics."""
    Kh = halve_intrinsics(K)
    Hr = Wr = img_size // 2
    C = feat.shape[1]
    r = render_rays(feat, dens, R, T, Kh, Hr, Wr, S, zmin, zmax, vol_size, render_depth)
    imgs = conv_rgb(r[..., :C].permute(0, 3, 1, 2).contiguous(), w, k_size=k_size, training=training)
    sil = F.interpolate(r[..., C:C + 1].permute(0, 3, 1, 2).contiguous(), size=[img_size] * 2,
                        mode="bilinear", align_corners=False)
    out = [imgs, sil]
    if render_depth:
        out.append(F.interpolate(r[..., C + 1:C + 2].permute(0, 3, 1, 2).contiguous(), size=[img_size] * 2,
                                 mode="bilinear", align_corners=False))
    if return_origin_proj:
        out.append(origin_projection(R, T, Kh))
    return tuple(out)


# --------------------------------------------------------------------------------------
# a8  orchestration
# --------------------------------------------------------------------------------------
def forward_pose3d_gt(sample, w, cfg, training=False):
    """FORGE_poseEstimator3D.forward, use_gt_pose=True, canonicalize=True
    (models/model_single_pose_estimator.py:26-138): t input views -> 2t rendered views
    [2v-fused volume x 3 cams, 3v-fused volume x 2 cams, mv-fused volume x t cams]."""
    imgs = sample["images"]
    b, t = imgs.shape[:2]
    feats = get_feat3D(imgs.reshape(b * t, *imgs.shape[2:]), w, training)
    C, D = feats.shape[1], feats.shape[2]
    feats = feats.reshape(b, t, C, D, D, D)
    camE = sample["cam_extrinsics_cv2_canonicalized"].repeat(1, 2, 1, 1).reshape(b * 2 * t, 4, 4)
    poses = sample["cam_poses_cv2_canonicalized"]
    K = sample["K_cv2"].repeat(1, 2, 1, 1).reshape(b * 2 * t, 3, 3)
    ft = rotate_world(feats, poses[:, :t], cfg.render.volume_size)
    f3 = fuse(ft[:, :3], w, training=training)
    f2 = fuse(ft[:, -2:], w, training=training)
    f32 = torch.cat([f3, f2], dim=0)
    d32 = density_head(f32, w, training=training)
    r32 = render_features_head(f32, w, training=training)
    fm = fuse(ft, w, training=training)
    dm = density_head(fm, w, training=training)
    rm = render_features_head(fm, w, training=training)
    rep = lambda v, n: v.unsqueeze(1).repeat(1, n, 1, 1, 1, 1)
    feat_all = torch.cat([rep(r32[b:], 3), rep(r32[:b], 2), rep(rm, t)], dim=1).reshape(b * 2 * t, *rm.shape[1:])
    dens_all = torch.cat([rep(d32[b:], 3), rep(d32[:b], 2), rep(dm, t)], dim=1).reshape(b * 2 * t, *dm.shape[1:])
    if cfg.dataset.name == "omniobject3d":
        dens_all = dens_all.clamp(min=0.0, max=1.0)
    return vol_render(feat_all, dens_all, camE[:, :3, :3], camE[:, :3, 3], K, w, cfg.dataset.img_size,
                      cfg.render.n_pts_per_ray, cfg.render.min_depth, cfg.render.max_depth,
                      cfg.render.volume_size, cfg.render.k_size, training=training)[:2]


def reconstruct_from_features(feats, poses, extrinsics, K, w, cfg, training=False, order_by_distance=False):
    """a2..a7 of models/model.py:127-145 on GIVEN per-view feature volumes feats [b,t,C,D,D,D]: rotate -> [order] -> fuse -> heads ->
    render the V = extrinsics.shape[1] cameras. D = 32 is what the encoder produces; D = 64 is the reference's large-grid path
    (models/rotate.py:115-117 -> 128^3 render volume, BASELINE configs[3]/[4])."""
    b = feats.shape[0]
    ft = rotate_world(feats, poses, cfg.render.volume_size)
    if order_by_distance:
        ft = chose_selected(ft, sequence_from_distance(poses[:, :, :3, 3]))
    fm = fuse(ft, w, training=training)
    dm = density_head(fm, w, training=training)
    rm = render_features_head(fm, w, training=training)
    V = extrinsics.shape[1]
    rep = lambda v: v.unsqueeze(1).repeat(1, V, 1, 1, 1, 1).reshape(b * V, *v.shape[1:])
    E = extrinsics.reshape(b * V, 4, 4)
    return vol_render(rep(rm), rep(dm), E[:, :3, :3], E[:, :3, 3], K.reshape(b * V, 3, 3), w,
                      cfg.dataset.img_size, cfg.render.n_pts_per_ray, cfg.render.min_depth,
                      cfg.render.max_depth, cfg.render.volume_size, cfg.render.k_size, training=training)[:2]


def forward_hot_path(images, poses, extrinsics, K, w, cfg, training=False, order_by_distance=False,
                     render_extrinsics=None, render_K=None):
    """The 5-in / V-out hot path a1..a7 of models/model.py:42-148 with poses GIVEN (pose
    estimators are out of scope): encode -> rotate -> [order] -> fuse -> heads -> render V views.
    images [b,t,3,H,W]; poses/extrinsics [b,t,4,4]; K [b,t,3,3]. By default renders the t input cameras."""
    b, t = images.shape[:2]
    feats = get_feat3D(images.reshape(b * t, *images.shape[2:]), w, training)
    C, D = feats.shape[1], feats.shape[2]
    feats = feats.reshape(b, t, C, D, D, D)
    return reconstruct_from_features(feats, poses, extrinsics if render_extrinsics is None else render_extrinsics,
                                     K if render_K is None else render_K, w, cfg, training, order_by_distance)


def psnr(a, b):
    """utils/eval_utils.py:8-12 with data_range=1: 10 log10(1/MSE)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)
