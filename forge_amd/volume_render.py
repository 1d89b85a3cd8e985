"""a6/a7 — differentiable volume renderer + neural up-sampler.

Mirror of the reference's models/volume_render.py::VolRender (:11-106): same constructor, same
`forward(camera_params, feature_3d, density_3d, render_depth=False, return_origin_proj=False)`
return orders (:77-88), same `proj_origin`, same state_dict keys (`conv_rgb.{0,1,3,4,6}.*`).
PyTorch3D is not a dependency: cameras_from_opencv_projection + NDCGridRaysampler + VolumeSampler +
EmissionAbsorptionRaymarcher (+ the README.md:26-33 depth patch) are one HIP launch
(forge_render_fwd) that never materialises ray points or sampled tensors.

Differences that are deliberate:
  * `camera_params['K']` is NOT mutated in place (the reference halves the caller's tensor, :50-51);
  * optional `view2vol`: when given, `feature_3d`/`density_3d` hold each scene volume ONCE and
    view v renders volume view2vol[v] — replaces the V-fold `repeat` of models/model.py:138-139.
    Without it the reference contract (one volume per view) applies unchanged.
"""
import torch
import torch.nn as nn

from . import _lib, convops as co, ops
from .fusion import affine_act_bwd, bn_act_rows, frozen_eval, hip_inference


class _ConvRgbFrozen(torch.autograd.Function):
    """conv_rgb + ReLU (models/volume_render.py:29-37,73) for frozen weights under autograd (pose refinement): forward = the fused
    inference launches of VolRender._conv_rgb_hip keeping the two intermediate activations; backward = data gradients only."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, vr):
        rgb, up, mid = vr._conv_rgb_hip(x, keep=True)
        ctx.vr, ctx.xshape = vr, x.shape
        ctx.save_for_backward(up, mid, rgb)
        return rgb

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, drgb):
        vr = ctx.vr
        up, mid, rgb = ctx.saved_tensors
        V, C, Hr, Wr = ctx.xshape
        p = vr._conv_rgb_packed_T()
        dev = up.device
        H2, W2 = 2 * Hr, 2 * Wr
        g2 = (V, 1, H2, W2)
        dr = drgb.permute(0, 2, 3, 1)
        dr = dr if dr.is_contiguous() else dr.contiguous()
        g = affine_act_bwd(dr, rgb.permute(0, 2, 3, 1), None, 0.0)                        # trailing ReLU
        dmid = torch.empty(V, H2, W2, 8, dtype=torch.float32, device=dev)
        co.direct_dgrad(g.reshape(V, 1, H2, W2, 3), p["w6"], dmid.reshape(V, 1, H2, W2, 8), g2, 8, 3, p["taps6"])
        g16 = torch.zeros(V, 1, H2, W2, 16, dtype=torch.float32, device=dev)
        affine_act_bwd(dmid.reshape(V, 1, H2, W2, 8), mid.reshape(V, 1, H2, W2, 8), p["bn4_scale"], 0.01, out=g16[..., :8])
        dup = torch.empty(V, 1, H2, W2, 16, dtype=torch.float32, device=dev)
        co.narrow_dgrad(g16, p["w3T"], dup, g2, p["taps3"])
        gu = affine_act_bwd(dup, up.reshape(V, 1, H2, W2, 16), p["bn1_scale"], 0.01)
        dx = torch.empty(V, 1, Hr, Wr, C, dtype=torch.float32, device=dev)
        co.conv_igemm(gu, 16, 16, None, 0, 0, p["ctT"], None, None, None, 1.0, None, None, None, dx, None, (V, 1, Hr, Wr), (1, H2, W2), C, C,
                      p["ct_taps"], istride=2, epilogue=co.EPI_BIAS)
        return dx.reshape(V, Hr, Wr, C).permute(0, 3, 1, 2), None


class VolRender(co.PackedModule):
    def __init__(self, config):
        super().__init__()
        self._rgb_cache = co.PackCache()
        self.img_size = config.dataset.img_size
        self.volume_physical_size = config.render.volume_size
        self.n_pts_per_ray = config.render.n_pts_per_ray
        self.min_depth = config.render.min_depth
        self.max_depth = config.render.max_depth
        self.k_size = config.render.k_size
        self.pad_size = self.k_size // 2
        self.conv_rgb = nn.Sequential(
            nn.ConvTranspose2d(16, 16, kernel_size=self.k_size + 1, stride=2, padding=self.pad_size),
            nn.BatchNorm2d(16),
            nn.LeakyReLU(inplace=True),
            nn.Conv2d(16, 8, kernel_size=self.k_size, stride=1, padding=self.pad_size),
            nn.BatchNorm2d(8),
            nn.LeakyReLU(inplace=True),
            nn.Conv2d(8, 3, kernel_size=self.k_size, stride=1, padding=self.pad_size),
        )

    # ray sharding (off by default): plain attributes, not parameters / buffers - the state_dict stays the reference's
    ray_shard = False              # True: the ray-march of every forward is split into row bands over the ranks of ray_shard_group
    ray_shard_group = None         # torch.distributed process group (None = the default group)
    ray_shard_reduce = "all"       # "all": d(volume) all-reduced (replicated encoder / fusion); "none": partials (dist.broadcast_from_owner)

    def _ray_shard_world(self, Hr):
        if not self.ray_shard:
            return 1
        import torch.distributed as tdist
        if not (tdist.is_available() and tdist.is_initialized()):
            return 1
        world = tdist.get_world_size(self.ray_shard_group)
        if world > 1 and Hr % world:
            raise ValueError("VolRender.ray_shard: %d image rows are not divisible by %d ranks" % (Hr, world))
        return world

    @staticmethod
    def _half_res_intrinsics(K):
        K = K.to(torch.float32) / 2.0          # copy; volume_render.py:50-51 without the in-place write
        K[:, -1, -1] = 1.0
        return K

    def _pack_cameras_hip(self, camera_params, device, want_origin):
        """Inference: camera packing + origin projection as ONE launch (forge_pack_cameras) on the possibly strided R / T / K views."""
        R, T, K = (camera_params[k].to(device=device, dtype=torch.float32) for k in ("R", "T", "K"))
        V = R.shape[0]
        cam = torch.empty(V, 16, dtype=torch.float32, device=device)
        origin = torch.empty(V, 2, dtype=torch.float32, device=device) if want_origin else None
        with torch.cuda.device(device):
            _lib.check(_lib.lib().forge_pack_cameras(_lib.ptr(R), R.stride(0), R.stride(1), R.stride(2), _lib.ptr(T), T.stride(0), T.stride(1),
                                                     _lib.ptr(K), K.stride(0), K.stride(1), K.stride(2), _lib.ptr(cam), _lib.ptr(origin), V,
                                                     _lib.current_stream()), "forge_pack_cameras")
        return cam, origin

    def _pack_cameras(self, camera_params, device):
        R = camera_params["R"].to(device=device, dtype=torch.float32)
        T = camera_params["T"].to(device=device, dtype=torch.float32)
        K = self._half_res_intrinsics(camera_params["K"].to(device))
        V = R.shape[0]
        cam = torch.cat([R.reshape(V, 9), T.reshape(V, 3), K[:, 0, 0:1], K[:, 1, 1:2], K[:, 0, 2:3], K[:, 1, 2:3]], dim=1)
        return cam, T, K

    @staticmethod
    def _origin_proj(T, K):
        """cameras.transform_points_screen(origin) (:77-79) == OpenCV projection of the world origin
        at half resolution."""
        return torch.stack([K[:, 0, 0] * T[:, 0] / T[:, 2] + K[:, 0, 2],
                            K[:, 1, 1] * T[:, 1] / T[:, 2] + K[:, 1, 2]], dim=-1)

    def forward(self, camera_params, feature_3d, density_3d, render_depth=False, return_origin_proj=False,
                view2vol=None):
        nvol, C, D, H, W = feature_3d.shape
        device = feature_3d.device
        origin = None
        if "packed" in camera_params:
            # extension (forge_amd/refine.py): cameras already packed as the ray-marcher wants them ([V,16]: R, T, fx fy cx cy at half
            # resolution, ops.pose_chain) together with the origin projection - no per-call camera algebra
            cam, origin = camera_params["packed"], camera_params.get("origin")
            if return_origin_proj and origin is None:
                raise ValueError("VolRender: packed cameras need their 'origin' projection for return_origin_proj")
        elif hip_inference(self, feature_3d) and not any(camera_params[k].requires_grad for k in ("R", "T", "K")):
            cam, origin = self._pack_cameras_hip(camera_params, device, return_origin_proj)
        else:
            cam, T, K = self._pack_cameras(camera_params, device)
        V = cam.shape[0]
        if view2vol is None:
            if V != nvol:
                raise ValueError("VolRender: %d cameras for %d volumes (pass view2vol to share volumes)" % (V, nvol))
            view2vol = torch.arange(V, dtype=torch.int32, device=device)
        else:
            view2vol = view2vol.to(device=device, dtype=torch.int32).contiguous()
        Hr = Wr = self.img_size // 2
        vox = self.volume_physical_size / D                       # :58 single_voxel_size
        half = (0.5 * (W - 1) * vox, 0.5 * (H - 1) * vox, 0.5 * (D - 1) * vox)
        shard_world = self._ray_shard_world(Hr)
        if shard_world > 1:
            # BASELINE configs[4] "per-ray sharding": this rank marches its band of image rows of every view, one all_gather assembles the
            # maps; backward = band backward + all-reduce (or reduce-to-owner) of d(volume) / d(cameras)  (forge_amd/dist.py)
            from . import dist as fdist
            outs = fdist.render_rays_sharded(feature_3d, density_3d, cam, view2vol, Hr, Wr, self.n_pts_per_ray, self.min_depth, self.max_depth, half,
                                             want_depth=render_depth, group=self.ray_shard_group, reduce=self.ray_shard_reduce)
        else:
            outs = ops.render_rays(feature_3d, density_3d, cam, view2vol, Hr, Wr, self.n_pts_per_ray,
                                   self.min_depth, self.max_depth, half, want_depth=render_depth)
        if hip_inference(self, outs[0]):
            rendered_imgs = self._conv_rgb_hip(outs[0])
        elif frozen_eval(self, outs[0]):
            rendered_imgs = _ConvRgbFrozen.apply(outs[0], self)      # refinement: fused forward, data-gradient-only backward
        else:
            rendered_imgs = self._conv_rgb_autograd_hip(outs[0])      # ops.render_rays has already refused non-HIP tensors
        rendered_silhouettes = ops.resize_bilinear(outs[1], self.img_size, self.img_size)          # :74, HIP kernel (forward + adjoint)
        result = [rendered_imgs, rendered_silhouettes]
        if render_depth:
            result.append(ops.resize_bilinear(outs[2], self.img_size, self.img_size))              # :69
        if return_origin_proj:
            result.append(origin if origin is not None else self._origin_proj(T, K))
        return tuple(result)

    def _conv_rgb_pack(self):
        """Packed / folded conv_rgb parameters of the fused inference launches (rebuilt when a source tensor changed)."""
        cr = self.conv_rgb
        src = [cr[0].weight, cr[0].bias, cr[3].weight, cr[3].bias, cr[6].weight, cr[6].bias] + \
              [t for bn in (cr[1], cr[4]) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]

        def build():
            w3, taps3 = co.pack_conv2d_weight(cr[3].weight)
            w6, taps6 = co.pack_conv2d_weight(cr[6].weight)
            return {"ct": co.convT_phases_merged(cr[0].weight, self.pad_size, 2), "ct_b": cr[0].bias.detach().contiguous(), "bn1": co.bn_affine(cr[1]),
                    "w3": w3, "taps3": taps3, "b3": cr[3].bias.detach().contiguous(), "bn4": co.bn_affine(cr[4]),
                    "w6": w6, "taps6": taps6, "b6": cr[6].bias.detach().contiguous()}
        return self._rgb_cache.get(src, build)

    def _conv_rgb_packed_T(self):
        p = self._conv_rgb_pack()               # re-pack if the cache was dropped between forward and backward (train()/eval(), .to(), ...)
        if "ctT" not in p:
            cr = self.conv_rgb
            k, pad = cr[0].weight.shape[-1], self.pad_size
            w0 = cr[0].weight.detach()                                                       # ConvTranspose2d weight [Cin, Cout, k, k]
            p.update({"ctT": w0.reshape(w0.shape[0], w0.shape[1], k * k).permute(2, 0, 1).contiguous(),      # [k*k][Cin][Cout]
                      "ct_taps": [(0, ky - pad, kx - pad) for ky in range(k) for kx in range(k)],
                      "w3T": co.pad_last(p["w3"].transpose(1, 2).contiguous(), 16), "bn1_scale": p["bn1"][0], "bn4_scale": p["bn4"][0]})
        return p

    def _conv_rgb_hip(self, x, keep=False):
        """conv_rgb + ReLU (models/volume_render.py:29-37,73) on the MFMA GEMM kernel: ConvTranspose2d(16,16,k+1,s2,p) as its 4
        output phases in one launch + folded BN + LeakyReLU, Conv2d(16,8,k)+BN+LeakyReLU, Conv2d(8,3,k)+ReLU. x [V,16,Hr,Wr] with
        channels-last memory (what the ray-marcher writes) -> [V,3,2Hr,2Wr] (channels-last memory)."""
        p = self._conv_rgb_pack()
        V, C, Hr, Wr = x.shape
        xr = x.permute(0, 2, 3, 1)
        xr = xr if xr.is_contiguous() else xr.contiguous()
        H2, W2 = 2 * Hr, 2 * Wr
        dev = x.device
        up = torch.empty(V, H2, W2, 16, dtype=torch.float32, device=dev)
        co.conv_igemm(xr, C, C, None, 0, 0, p["ct"][1], p["ct_b"], p["bn1"][0], p["bn1"][1], 0.01, None, None, None, up, None,
                      (V, 1, Hr, Wr), (1, Hr, Wr), 16, 16, p["ct"][0], out_grid=(1, H2, W2), ostride=2, phase=(-1, -1, -1),
                      epilogue=co.EPI_AFFINE_ACT)
        g2, ig2 = (V, 1, H2, W2), (1, H2, W2)
        mid = torch.empty(V, H2, W2, 8, dtype=torch.float32, device=dev)
        co.conv_igemm(up, 16, 16, None, 0, 0, p["w3"], p["b3"], p["bn4"][0], p["bn4"][1], 0.01, None, None, None, mid, None,
                      g2, ig2, 8, 8, p["taps3"], epilogue=co.EPI_AFFINE_ACT)
        rgb = torch.empty(V, H2, W2, 3, dtype=torch.float32, device=dev)
        co.conv_direct(mid, 8, p["w6"], p["b6"], 0.0, rgb, g2, 8, 3, p["taps6"])              # Conv2d(8, 3, k) + ReLU on the vector ALUs
        if keep:
            return rgb.permute(0, 3, 1, 2), up, mid
        return rgb.permute(0, 3, 1, 2)

    def _conv_rgb_autograd_hip(self, x):
        """conv_rgb + ReLU with an autograd graph (training / pose refinement): the transposed convolution and Conv2d(16, 8, k) on the
        narrow-N GEMM kernel (forward, data gradient) and the narrow weight-gradient kernels, Conv2d(8, 3, k) on the direct
        kernels; BatchNorm2d (batch statistics / SyncBN) and the activations stay torch ops on the same NHWC memory."""
        cr = self.conv_rgb
        rows = x.permute(0, 2, 3, 1)
        rows = rows if rows.is_contiguous() else rows.contiguous()
        V, Hr, Wr, C = rows.shape

        bn_act = lambda bn, r: bn_act_rows(bn, r, 0.01)             # r [V,H,W,C]: BatchNorm2d + LeakyReLU, one fused pass each way in train mode

        up = co.convT_s2_rows(rows.reshape(V, 1, Hr, Wr, C), cr[0].weight, cr[0].bias, self.pad_size, 2).reshape(V, 2 * Hr, 2 * Wr, -1)
        mid = bn_act(cr[4], co.conv2d_rows_any(bn_act(cr[1], up), cr[3].weight, cr[3].bias))
        rgb = torch.relu(co.conv2d_rows_any(mid, cr[6].weight, cr[6].bias))
        return rgb.permute(0, 3, 1, 2)

    def proj_origin(self, camera_params, device):
        """models/volume_render.py:91-103"""
        _, T, K = self._pack_cameras(camera_params, device)
        return self._origin_proj(T, K)
