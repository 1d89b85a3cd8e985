"""Analytic known-answer cases for the ray-marcher (row a6), derived from the DOCUMENTED contracts of the PyTorch3D pieces
models/volume_render.py:18-24,53-63 wires together — not from oracle/shims, not from the oracle's closed form, and without any
grid_sample / trilinear-gather code:

  * cameras_from_opencv_projection: the camera projects world points exactly as OpenCV does, X_c = R X + t, (u, v) = (fx x/z + cx,
    fy y/z + cy) in pixels of the (img/2)^2 render target;
  * NDCGridRaysampler: one ray per pixel through the pixel CENTRE (w + 0.5, h + 0.5), n_pts_per_ray depths linspace(min, max) measured
    along the camera z axis;
  * Volumes(voxel_size = vol / D): voxel (iz, iy, ix) of a [D, H, W] grid has its centre at world
    ((ix - (W-1)/2) s, (iy - (H-1)/2) s, (iz - (D-1)/2) s), s = vol / D (x <-> W, y <-> H, z <-> D); VolumeSampler interpolates
    linearly between voxel centres and pads with zeros, so a field that is ONE voxel of value d0 reads d0 * prod_axis max(0, 1 - |u_axis -
    i_axis|) (u = continuous voxel coordinate) and a CONSTANT field reads d0 * prod_axis tent(u_axis) with tent = 1 between the first
    and last centre, falling linearly to 0 one voxel outside (features are sampled the same way as densities);
  * EmissionAbsorptionRaymarcher: w_s = d_s prod_{j<s} (1 - d_j), features sum w_s f_s, opacity 1 - prod (1 - d_s), depth sum w_s z_s
    (README.md:26-33 patch).

Every expected value below is evaluated in float64 numpy / python loops from those statements. Cases:
  slab_cubic        constant density/feature, canonical camera (dataset/kubric.py:100-104), the ray through the principal point: depth
                    sample placement, fringe handling, compositing order, depth channel
  slab_aniso        [8, 12, 20] grid seen along -x / along -y: in-volume sample counts differ per axis -> catches x<->W / y<->H swaps
  impulse           one voxel at an off-centre, all-different index of a non-cubic grid, rotated camera with fx != fy, cx != cy:
                    hit pixel = floor(OpenCV projection of the voxel centre); full opacity / feature image from the tent-product formula
  impulse_on_sample the same with a depth sample placed exactly on the voxel centre: opacity == d0 at the hit pixel
  impulse_tall      a TALL render target (Hr = 44 > Wr = 26; every other case is wider than tall or square): NDCGridRaysampler's second
                    non-square branch (range_y = H / W, range_x = 1) and cameras_from_opencv_projection's min(H, W) scale with H != W
  slab_tall         a constant slab on the tall target, pixels in the corners of the long axis

All cases with Hr != Wr also pin the PyTorch3D restatement used for the golden import (oracle/shims/pytorch3d: test_oracle_golden.py::
test_pytorch3d_shim_reproduces_analytic_kats_on_non_square_targets), in both of its non-square branches.
"""
import math

import numpy as np


def _tent_const(u, n):
    """Linear interpolation of a constant-1 field on n voxel centres with zero padding, at continuous coordinate u."""
    if 0.0 <= u <= n - 1:
        return 1.0
    if -1.0 < u < 0.0:
        return 1.0 + u
    if n - 1 < u < n:
        return n - u
    return 0.0


def _voxel_coord(p, dims, s):
    """world point (x, y, z) -> continuous voxel coordinates (ux, uy, uz) of a [D, H, W] grid with voxel size s."""
    D, H, W = dims
    return (p[0] / s + 0.5 * (W - 1), p[1] / s + 0.5 * (H - 1), p[2] / s + 0.5 * (D - 1))


def _composite(d, f, z):
    """EA ray-marcher on python lists: densities d [S], features f [S][C], depths z [S] -> (feat [C], opacity, depth)."""
    T = 1.0
    C = len(f[0])
    feat, depth = [0.0] * C, 0.0
    for ds, fs, zs in zip(d, f, z):
        w = ds * T
        for c in range(C):
            feat[c] += w * fs[c]
        depth += w * zs
        T *= (1.0 - ds)
    return feat, 1.0 - T, depth


def _camera(R, t, fx, fy, cx, cy):
    return {"R": np.asarray(R, np.float64), "T": np.asarray(t, np.float64), "fx": fx, "fy": fy, "cx": cx, "cy": cy}


def _pixel_ray(cam, h, w):
    """Origin and direction (camera-z component 1) of the ray through the centre of pixel (h, w), in world coordinates: the world
    points X with R X + t = z * ((w + .5 - cx) / fx, (h + .5 - cy) / fy, 1)."""
    Rt = cam["R"].T
    d_cam = np.array([(w + 0.5 - cam["cx"]) / cam["fx"], (h + 0.5 - cam["cy"]) / cam["fy"], 1.0])
    return -Rt @ cam["T"], Rt @ d_cam


def slab_case(dims, vol, d0, fvals, cam, pixels, S, zmin, zmax):
    """Constant density d0 / constant features fvals over a [D,H,W] grid; expected (feat, opacity, depth) at the listed pixels."""
    s = vol / dims[0]
    zs = np.linspace(zmin, zmax, S)
    exp = {}
    for (h, w) in pixels:
        o, d = _pixel_ray(cam, h, w)
        dens, feats = [], []
        for z in zs:
            ux, uy, uz = _voxel_coord(o + d * z, dims, s)
            tent = _tent_const(ux, dims[2]) * _tent_const(uy, dims[1]) * _tent_const(uz, dims[0])
            dens.append(d0 * tent)
            feats.append([f * tent for f in fvals])          # the feature field is sampled exactly like the density (zero padding)
        exp[(h, w)] = _composite(dens, feats, list(zs))
    return exp


def impulse_case(dims, vol, idx, d0, fvals, cam, Hr, Wr, S, zmin, zmax):
    """One voxel idx = (iz, iy, ix) holds density d0 and features fvals, everything else 0 (features are zero elsewhere too, and the
    sampled feature is interpolated like the density). Returns (proj (u, v), feat [Hr,Wr,C], opacity [Hr,Wr], depth [Hr,Wr])."""
    D, H, W = dims
    s = vol / D
    iz, iy, ix = idx
    centre = np.array([(ix - 0.5 * (W - 1)) * s, (iy - 0.5 * (H - 1)) * s, (iz - 0.5 * (D - 1)) * s])
    pc = cam["R"] @ centre + cam["T"]
    proj = (cam["fx"] * pc[0] / pc[2] + cam["cx"], cam["fy"] * pc[1] / pc[2] + cam["cy"])
    zs = np.linspace(zmin, zmax, S)
    Rt = cam["R"].T
    o = -Rt @ cam["T"]
    hh, ww = np.meshgrid(np.arange(Hr) + 0.5, np.arange(Wr) + 0.5, indexing="ij")
    d_cam = np.stack([(ww - cam["cx"]) / cam["fx"], (hh - cam["cy"]) / cam["fy"], np.ones_like(ww)], axis=-1)      # [Hr,Wr,3]
    d_w = d_cam @ Rt.T                                                                                             # row-vector form of Rt @ d
    P = o[None, None, None, :] + d_w[:, :, None, :] * zs[None, None, :, None]                                      # [Hr,Wr,S,3]
    ux, uy, uz = P[..., 0] / s + 0.5 * (W - 1), P[..., 1] / s + 0.5 * (H - 1), P[..., 2] / s + 0.5 * (D - 1)
    wgt = np.clip(1 - np.abs(ux - ix), 0, None) * np.clip(1 - np.abs(uy - iy), 0, None) * np.clip(1 - np.abs(uz - iz), 0, None)
    dens = d0 * wgt                                                                                                # [Hr,Wr,S]
    T = np.cumprod(np.concatenate([np.ones_like(dens[..., :1]), 1 - dens[..., :-1]], axis=-1), axis=-1)
    w_s = dens * T
    fv = np.asarray(fvals, np.float64)
    feat = (w_s * wgt).sum(-1)[..., None] * fv[None, None, :]                                                      # sampled feature = fvals * wgt
    return proj, feat, 1 - np.prod(1 - dens, axis=-1), (w_s * zs).sum(-1)


# ---------------------------------------------------------------------------------------------------------------------
# the concrete cases (inputs as numpy arrays in the layouts the oracle / the HIP op take)
# ---------------------------------------------------------------------------------------------------------------------
def cases():
    out = []
    # -- slab_cubic: canonical camera at z = -1.5 looking along +z, principal point on the centre of pixel (10, 12)
    D = 16
    cam = _camera(np.eye(3), [0.0, 0.0, 1.5], 44.0, 44.0, 12.5, 10.5)
    fv = [0.7, -1.3, 2.0, 0.25]
    px = [(10, 12), (10, 13), (3, 20), (0, 0)]
    out.append({"name": "slab_cubic", "dims": (D, D, D), "vol": 1.0, "dens": np.full((D, D, D), 0.3), "feat": np.tile(np.array(fv)[:, None, None, None], (1, D, D, D)),
                "cam": cam, "Hr": 21, "Wr": 25, "S": 40, "zmin": 0.5, "zmax": 2.0, "expect_pixels": slab_case((D, D, D), 1.0, 0.3, fv, cam, px, 40, 0.5, 2.0)})
    # -- slab_aniso: [D,H,W] = [8,12,20], voxel 1/8: extents x 2.375, y 1.375, z 0.875 (centre to centre)
    dims = (8, 12, 20)
    fv = [1.0, 0.5, -0.5, 2.0]
    cam_x = _camera([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], [0.0, 0.0, 3.0], 40.0, 40.0, 8.5, 8.5)      # at (3,0,0) looking along -x
    cam_y = _camera([[1, 0, 0], [0, 0, 1], [0, -1, 0]], [0.0, 0.0, 3.0], 40.0, 40.0, 8.5, 8.5)      # at (0,3,0) looking along -y
    for nm, cam in (("slab_aniso_x", cam_x), ("slab_aniso_y", cam_y)):
        px = [(8, 8), (8, 9), (2, 5)]
        out.append({"name": nm, "dims": dims, "vol": 1.0, "dens": np.full(dims, 0.05), "feat": np.tile(np.array(fv)[:, None, None, None], (1,) + dims),
                    "cam": cam, "Hr": 17, "Wr": 17, "S": 96, "zmin": 1.0, "zmax": 5.0, "expect_pixels": slab_case(dims, 1.0, 0.05, fv, cam, px, 96, 1.0, 5.0)})
    # -- impulse: voxel (iz, iy, ix) = (2, 9, 13) of a [8,12,20] grid, camera rotated about y by 35 deg and about x by -20 deg, fx != fy, cx != cy
    a, b = math.radians(35.0), math.radians(-20.0)
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    R = Rx @ Ry
    cam = _camera(R, [0.1, -0.05, 2.5], 90.0, 70.0, 27.0, 19.0)
    idx, d0, fv = (2, 9, 13), 0.8, [1.5, -2.0, 0.5, 1.0]
    dens = np.zeros(dims)
    dens[idx] = d0
    feat = np.zeros((4,) + dims)
    feat[(slice(None),) + idx] = fv
    proj, ef, eo, ed = impulse_case(dims, 1.0, idx, d0, fv, cam, 40, 48, 160, 1.0, 4.0)
    out.append({"name": "impulse", "dims": dims, "vol": 1.0, "dens": dens, "feat": feat, "cam": cam, "Hr": 40, "Wr": 48, "S": 160, "zmin": 1.0, "zmax": 4.0,
                "proj": proj, "expect_feat": ef, "expect_opacity": eo, "expect_depth": ed})
    # -- impulse_on_sample: canonical-style camera; pixel (h0, w0) centre and depth sample 0 exactly on the voxel centre -> opacity d0 there
    D = 16
    s = 1.0 / D
    idx = (11, 4, 9)
    c = np.array([(idx[2] - 7.5) * s, (idx[1] - 7.5) * s, (idx[0] - 7.5) * s])
    tz = 1.5
    zc = c[2] + tz
    fx, fy, h0, w0 = 50.0, 60.0, 6, 17
    cam = _camera(np.eye(3), [0.0, 0.0, tz], fx, fy, w0 + 0.5 - fx * c[0] / zc, h0 + 0.5 - fy * c[1] / zc)
    dens = np.zeros((D, D, D))
    dens[idx] = 0.6
    fv = [2.0, 1.0, -1.0, 0.5]
    feat = np.zeros((4, D, D, D))
    feat[(slice(None),) + idx] = fv
    proj, ef, eo, ed = impulse_case((D, D, D), 1.0, idx, 0.6, fv, cam, 16, 24, 2, zc, zc + 0.5)
    out.append({"name": "impulse_on_sample", "dims": (D, D, D), "vol": 1.0, "dens": dens, "feat": feat, "cam": cam, "Hr": 16, "Wr": 24, "S": 2, "zmin": zc,
                "zmax": zc + 0.5, "proj": proj, "expect_feat": ef, "expect_opacity": eo, "expect_depth": ed, "hit": (h0, w0), "hit_opacity": 0.6})
    # -- impulse_tall / slab_tall: Hr > Wr (PyTorch3D's NDC range is 1 along the SHORT side: range_y = Hr / Wr here)
    a, b = math.radians(-25.0), math.radians(15.0)
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    cam = _camera(Rx @ Ry, [-0.05, 0.12, 2.2], 60.0, 85.0, 11.0, 24.5)
    dims = (10, 14, 6)
    idx, d0, fv = (7, 3, 4), 0.9, [0.5, 1.0, -1.5, 2.5]
    dens = np.zeros(dims)
    dens[idx] = d0
    feat = np.zeros((4,) + dims)
    feat[(slice(None),) + idx] = fv
    proj, ef, eo, ed = impulse_case(dims, 1.0, idx, d0, fv, cam, 44, 26, 120, 1.0, 3.5)
    out.append({"name": "impulse_tall", "dims": dims, "vol": 1.0, "dens": dens, "feat": feat, "cam": cam, "Hr": 44, "Wr": 26, "S": 120, "zmin": 1.0, "zmax": 3.5,
                "proj": proj, "expect_feat": ef, "expect_opacity": eo, "expect_depth": ed})
    fv = [0.3, -0.6, 1.2, 0.9]
    cam = _camera(np.eye(3), [0.0, 0.0, 1.5], 30.0, 36.0, 13.0, 22.0)
    px = [(22, 13), (0, 0), (43, 25), (43, 0), (5, 20), (40, 12)]
    out.append({"name": "slab_tall", "dims": (D, D, D), "vol": 1.0, "dens": np.full((D, D, D), 0.2), "feat": np.tile(np.array(fv)[:, None, None, None], (1, D, D, D)),
                "cam": cam, "Hr": 44, "Wr": 26, "S": 48, "zmin": 0.5, "zmax": 2.5, "expect_pixels": slab_case((D, D, D), 1.0, 0.2, fv, cam, px, 48, 0.5, 2.5)})
    return out


def check(case, got, tol=2e-5):
    """got [Hr,Wr,C+2] float array = (features, opacity, depth) of the renderer under test. Raises AssertionError with the case name."""
    nm = case["name"]
    C = case["feat"].shape[0]
    got = np.asarray(got, np.float64)
    assert got.shape == (case["Hr"], case["Wr"], C + 2), (nm, got.shape)
    if "expect_pixels" in case:
        for (h, w), (ef, eo, ed) in case["expect_pixels"].items():
            assert np.abs(got[h, w, :C] - np.array(ef)).max() < tol * max(1.0, np.abs(ef).max()), (nm, (h, w), got[h, w, :C], ef)
            assert abs(got[h, w, C] - eo) < tol, (nm, (h, w), got[h, w, C], eo)
            assert abs(got[h, w, C + 1] - ed) < tol * max(1.0, abs(ed)), (nm, (h, w), got[h, w, C + 1], ed)
        return
    u, v = case["proj"]
    op = got[..., C]
    assert op.max() > 0, nm
    hmax, wmax = np.unravel_index(np.argmax(op), op.shape)
    # the brightest pixel is the one containing the OpenCV projection of the voxel centre (or an immediate neighbour when the
    # projection falls near a pixel border: the blob is a few pixels wide and symmetric about the projection)
    assert abs((wmax + 0.5) - u) <= 1.0 and abs((hmax + 0.5) - v) <= 1.0, (nm, (hmax, wmax), (v, u))
    ws, hs = np.arange(op.shape[1]) + 0.5, np.arange(op.shape[0]) + 0.5
    cu, cv = (op.sum(0) * ws).sum() / op.sum(), (op.sum(1) * hs).sum() / op.sum()
    assert abs(cu - u) < 0.5 and abs(cv - v) < 0.5, (nm, "centroid", (cv, cu), (v, u))
    assert np.abs(op - case["expect_opacity"]).max() < tol, (nm, "opacity", np.abs(op - case["expect_opacity"]).max())
    assert np.abs(got[..., :C] - case["expect_feat"]).max() < tol * max(1.0, np.abs(case["expect_feat"]).max()), (nm, "features")
    assert np.abs(got[..., C + 1] - case["expect_depth"]).max() < tol * max(1.0, case["zmax"]), (nm, "depth")
    if "hit" in case:
        assert abs(op[case["hit"]] - case["hit_opacity"]) < tol, (nm, op[case["hit"]])
        assert (op > 0).sum() < 0.25 * op.size, nm
