"""a1/a5 — 2D->3D feature lift, fusion wrapper and the density / render-feature heads.

Mirror of the reference's models/encoder.py (Encoder3D :8-68, get_resnet50 :71-78): same attribute
names (`feature_extraction`, `features_head`, `density_head`, `conv1`, `fusion_feature`), same
methods (`get_feat3D`, `get_density3D`, `get_render_features`, `fuse`) and the same state_dict keys
(SURVEY.md Appendix B). torchvision is not a dependency: the ResNet-50 trunk is built here with
torchvision's module names so published checkpoints load with strict=True.
"""
import torch
import torch.nn as nn

from . import _lib, convops as co
from .fusion import ConvGRU_3D, affine_act_bwd, bn_act_rows, frozen_eval, hip_inference, require_hip_input


class _Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck (stride on the 3x3 conv2), torchvision child names."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        identity = x if self.downsample is None else self.downsample(x)
        return self.relu(out + identity)


def _make_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes * 4:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * 4))
    layers = [_Bottleneck(inplanes, planes, stride, downsample)]
    layers += [_Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def get_resnet50():
    """models/encoder.py:71-78: torchvision resnet50 children[:-2] as nn.Sequential (indices 0..7 =
    conv1, bn1, relu, maxpool, layer1..4) with layer3[0] / layer4[0] conv2 + downsample stride set to 1,
    i.e. total stride 8. Weights are randomly initialised (kaiming, as torchvision does); the
    ImageNet checkpoint the reference downloads arrives through load_state_dict like any other."""
    feature = nn.Sequential(
        nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False),
        nn.BatchNorm2d(64),
        nn.ReLU(inplace=True),
        nn.MaxPool2d(3, stride=2, padding=1),
        _make_layer(64, 64, 3, 1),
        _make_layer(256, 128, 4, 2),
        _make_layer(512, 256, 6, 2),
        _make_layer(1024, 512, 3, 2),
    )
    for m in feature.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    feature[7][0].conv2.stride = (1, 1)
    feature[7][0].downsample[0].stride = (1, 1)
    feature[6][0].conv2.stride = (1, 1)
    feature[6][0].downsample[0].stride = (1, 1)
    return feature


def load_imagenet_trunk(feature_extraction, state_dict, strict=True):
    """Load a torchvision ResNet-50 state_dict (`resnet50(pretrained=True)`, what models/encoder.py:72 downloads) into the trunk built by
    get_resnet50(): torchvision key -> nn.Sequential index  conv1 -> 0, bn1 -> 1, layerN -> N + 3; `fc.*` is dropped (children[:-2]).
    Needed only for from-scratch training (stage 1.1 of the reference starts from ImageNet weights); a released FORGE checkpoint already
    carries the trunk under `encoder_3d.feature_extraction.*`. Shapes are validated; returns load_state_dict's result."""
    remap = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}
    own = feature_extraction.state_dict()
    out = {}
    for k, v in state_dict.items():
        head, _, rest = k.partition(".")
        if head == "fc":
            continue
        if head not in remap:
            raise KeyError("load_imagenet_trunk: unexpected torchvision ResNet-50 key %r" % k)
        nk = remap[head] + "." + rest
        if nk in own and tuple(own[nk].shape) != tuple(v.shape):
            raise ValueError("load_imagenet_trunk: %s has shape %s, the trunk expects %s" % (k, tuple(v.shape), tuple(own[nk].shape)))
        out[nk] = v
    return feature_extraction.load_state_dict(out, strict=strict)


@_lib.on_tensor_device
def resnet_rows_autograd(conv0, bn0, pool, stages, img, slope=0.0):
    """A torchvision-style bottleneck ResNet (stem conv0 / bn0 / pool + `stages` = sequences of blocks with conv1..3, bn1..3, downsample) with
    an autograd graph on the HIP kernels, activations as NHWC rows; returns the output rows of every stage. `slope`: 0 = ReLU (the encoder's
    trunk, models/encoder.py:71-78), 0.01 = the LeakyReLU bottlenecks of the 2-D pose estimator's FPN (models/pose_estimator_2d.py:237-275).
    Stem: patch gather + one-tap GEMM (the image needs no gradient), max-pool by torch; bottlenecks: 1x1 / 3x3 (stride 1 or 2) / 1x1
    convolutions through convops.conv2d_rows, every BatchNorm's batch statistics from its convolution's GEMM epilogue."""
    N, Ci, Hi, Wi = img.shape
    kh, kw = conv0.kernel_size
    s0, p0 = conv0.stride[0], conv0.padding[0]
    Kp = ((kh * kw * Ci + 31) // 32) * 32
    Hc, Wc = (Hi + 2 * p0 - kh) // s0 + 1, (Wi + 2 * p0 - kw) // s0 + 1
    patches = torch.empty(N, 1, Hc, Wc, Kp, dtype=torch.float32, device=img.device)
    _lib.check(_lib.lib().forge_im2col_nchw(_lib.ptr(img.detach().contiguous()), _lib.ptr(patches), N, Ci, Hi, Wi, kh, kw, s0, p0, Kp,
                                            _lib.current_stream()), "forge_im2col_nchw")
    w0 = torch.nn.functional.pad(conv0.weight.permute(0, 2, 3, 1).reshape(conv0.out_channels, -1), (0, Kp - kh * kw * Ci))[None]
    x, st0 = co.conv_taps_rows(patches, None, w0, None, [(0, 0, 0)], want_stats=True)
    x = bn_act_rows(bn0, x.reshape(N, Hc, Wc, conv0.out_channels), slope, stats=st0)
    x = pool(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    outs = []
    for stage in stages:
        for blk in stage:
            # conv1 hands its input through: the gradients of the identity / downsample path are added inside conv1's data-gradient GEMM
            y1, idn, st1 = co.conv1x1_rows_skip(x, blk.conv1.weight)
            out = bn_act_rows(blk.bn1, y1, slope, stats=st1)              # the statistics of every BatchNorm below come from its convolution's GEMM epilogue
            y2, st2 = co.conv2d_rows(out, blk.conv2.weight, None, stride=blk.conv2.stride[0], want_stats=True)
            out = bn_act_rows(blk.bn2, y2, slope, stats=st2)
            if blk.downsample is not None:
                yd, std = co.conv2d_rows(idn, blk.downsample[0].weight, None, stride=blk.downsample[0].stride[0], want_stats=True)
                idn = bn_act_rows(blk.downsample[1], yd, 1.0, stats=std)
            # act(bn3(conv3) + identity) in bn3's apply pass (and its mask / d identity in bn3's backward apply pass)
            y3, st3 = co.conv2d_rows(out, blk.conv3.weight, None, want_stats=True)
            x = bn_act_rows(blk.bn3, y3, slope, residual=idn, stats=st3)
        outs.append(x)
    return outs


class _HeadsFrozen(torch.autograd.Function):
    """Both heads (models/encoder.py:16-34) for frozen weights under autograd (pose refinement): forward = the fused inference launches
    of Encoder3D._heads_hip (merged transposed convolutions, BatchNorm / activations in the epilogues) keeping the two intermediate
    activations; backward = data gradients only - activation / BatchNorm-scale masks (forge_affine_act_bwd), the narrow layers on the
    narrow-N and direct kernels, the transposed convolutions as ONE stride-2 gather GEMM over their 64 taps."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, z, enc):
        feat, dens, up, d8 = enc._heads_hip(z, "both", keep=True)
        ctx.enc, ctx.zshape = enc, z.shape
        ctx.save_for_backward(up, d8, dens)            # dens is an output of this node: autograd records it without a reference cycle
        return feat, dens

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dfeat, ddens):
        enc = ctx.enc
        up, d8, dens = ctx.saved_tensors
        n, C, D, H, W = ctx.zshape
        p = enc._heads_packed_T()
        dev = up.device
        D2, H2, W2 = 2 * D, 2 * H, 2 * W
        g2 = (n, D2, H2, W2)
        M2 = n * D2 * H2 * W2
        rows = lambda t_: (lambda r_: r_ if r_.is_contiguous() else r_.contiguous())(t_.permute(0, 2, 3, 4, 1))
        dup = torch.empty(n, D2, H2, W2, 64, dtype=torch.float32, device=dev)
        # features head: Conv3d(32,16)+BN  (no activation)
        gf = affine_act_bwd(rows(dfeat), rows(dfeat), p["f4_scale"], 1.0)
        co.narrow_dgrad(gf, p["f3_wT"], dup[..., :32], g2, co.TAPS_3x3x3)
        # density head: Conv3d(8,1)+ReLU <- Conv3d(32,8)+BN+LReLU
        drows, densr = rows(ddens), dens.permute(0, 2, 3, 4, 1)
        gd = affine_act_bwd(drows, densr, None, 0.0)
        dd8 = torch.empty(n, D2, H2, W2, 8, dtype=torch.float32, device=dev)
        co.direct_dgrad(gd, p["d6_w"], dd8, g2, 8, 1, co.TAPS_3x3x3)
        gd16 = torch.zeros(n, D2, H2, W2, 16, dtype=torch.float32, device=dev)
        affine_act_bwd(dd8, d8, p["d4_scale"], 0.01, out=gd16[..., :8])
        co.narrow_dgrad(gd16, p["d3_wT"], dup[..., 32:], g2, co.TAPS_3x3x3)
        # both transposed convolutions (+BN+LReLU) at once: dz[v] = sum_k g[2v - 1 + k] W[:, :, k]
        gu = affine_act_bwd(dup, up, p["ct_scale"], 0.01)
        dz = torch.empty(n, D, H, W, C, dtype=torch.float32, device=dev)
        co.conv_igemm(gu, 64, 64, None, 0, 0, p["ct_wT"], None, None, None, 1.0, None, None, None, dz, None, (n, D, H, W), (D2, H2, W2), C, C,
                      p["ct_taps"], istride=2, epilogue=co.EPI_BIAS)
        return dz.permute(0, 4, 1, 2, 3), None


class Encoder3D(co.PackedModule):
    """models/encoder.py:8-68."""

    def __init__(self, config):
        super().__init__()
        self._trunk_cache, self._stem_cache, self._c1_cache, self._heads_cache = co.PackCache(), co.PackCache(), co.PackCache(), co.PackCache()
        self.feature_extraction = get_resnet50()
        self.features_head = nn.Sequential(
            nn.ConvTranspose3d(128, 32, 4, stride=2, padding=1),
            nn.BatchNorm3d(32),
            nn.LeakyReLU(inplace=True),
            nn.Conv3d(32, 16, 3, padding=1),
            nn.BatchNorm3d(16),
        )
        self.density_head = nn.Sequential(
            nn.ConvTranspose3d(128, 32, 4, stride=2, padding=1),
            nn.BatchNorm3d(32),
            nn.LeakyReLU(inplace=True),
            nn.Conv3d(32, 8, 3, padding=1),
            nn.BatchNorm3d(8),
            nn.LeakyReLU(inplace=True),
            nn.Conv3d(8, 1, 3, padding=1),
            nn.ReLU(inplace=True),
        )
        self.conv1 = nn.Sequential(
            nn.Conv3d(64, 128, 3, padding=1),
            nn.BatchNorm3d(128),
            nn.LeakyReLU(inplace=True),
        )
        self.fusion_feature = ConvGRU_3D(config, n_layers=1, input_size=128, hidden_size=128)

    def get_feat3D(self, img):
        """[N,3,H,W] -> [N,128,32,H/8,W/8]; the 2048 trunk channels are re-read as 64 ch x 32 depth
        (channel c = c3d*32 + z, models/encoder.py:49)."""
        if hip_inference(self, img):
            return self._conv1_hip(self._trunk_hip(img))
        require_hip_input("Encoder3D.get_feat3D", img)
        # training / refinement: every convolution (trunk, conv1) forward + dgrad on the MFMA GEMM, wgrad on the wgrad kernels;
        # BatchNorm (batch statistics / SyncBN) and activations are torch ops on the same channels-last tensors
        z = self._trunk_autograd_hip(img)                              # [N,H,W,2048] NHWC rows
        N, H, W, _ = z.shape
        rows = z.reshape(N, H, W, 64, 32).permute(0, 4, 1, 2, 3).contiguous()     # view(-1,64,32,H,W) as rows [N,32,H,W,64]
        y = co.conv3x3x3_rows(rows, None, self.conv1[0].weight, self.conv1[0].bias)
        return bn_act_rows(self.conv1[1], y, self.conv1[2].negative_slope).permute(0, 4, 1, 2, 3)

    def get_density3D(self, z_3d):
        """models/encoder.py:53-54. Callers that need both heads of the same volume should use heads() (one shared launch)."""
        if hip_inference(self, z_3d):
            return self._heads_hip(z_3d, "density")[1]
        require_hip_input("Encoder3D.get_density3D", z_3d)
        return self._head_autograd_hip(self.density_head, z_3d)

    def get_render_features(self, x):
        """models/encoder.py:56-57."""
        if hip_inference(self, x):
            return self._heads_hip(x, "features")[0]
        require_hip_input("Encoder3D.get_render_features", x)
        return self._head_autograd_hip(self.features_head, x)

    def heads(self, z_3d):
        """(get_render_features(z), get_density3D(z)) of the same fused volume. In inference both heads' transposed convolutions run as
        ONE N = 64 launch; nothing is memoised between calls (the model classes call this instead of the two getters)."""
        if hip_inference(self, z_3d):
            return self._heads_hip(z_3d, "both")
        if frozen_eval(self, z_3d):
            return _HeadsFrozen.apply(z_3d, self)                        # refinement: fused forward, data-gradient-only backward
        return self.get_render_features(z_3d), self.get_density3D(z_3d)

    def fuse(self, x, skip_dx0=False, const0=None):
        """x [b,t,c,d,h,w] -> [b,c,d,h,w] (models/encoder.py:59-63). skip_dx0 (frozen weights under autograd only): the caller needs no gradient
        for view 0 of the sequence - the fixed, un-warped reference view of a pose-refinement problem (ConvGRU_3D.fuse_frozen_hip); const0: that
        view is also the SAME in every call made with this dict (its input-half products are computed once and kept there)."""
        if hip_inference(self, x):
            return self.fusion_feature.fuse_hip(x)
        require_hip_input("Encoder3D.fuse", x)
        if frozen_eval(self.fusion_feature, x) and self.fusion_feature.n_layers == 1:
            return self.fusion_feature.fuse_frozen_hip(x, skip_dx0=skip_dx0, const0=const0)     # refinement: fused forward, data-gradient-only backward
        return self.fusion_feature.fuse_autograd_hip(x)                 # training: HIP convs with autograd

    def fuse_groups(self, x, groups):
        """[self.fuse(x[:, g]) for g in groups], sharing the per-view work between the groups: the input halves of the GRU convolutions
        are computed once per view (ConvGRU_3D.fuse_groups_hip in inference, fuse_groups_autograd_hip in training)."""
        if self.fusion_feature.n_layers == 1 and len(groups) > 1:
            require_hip_input("Encoder3D.fuse_groups", x)
            if hip_inference(self, x):
                return self.fusion_feature.fuse_groups_hip(x, groups)
            if not frozen_eval(self.fusion_feature, x):
                return self.fusion_feature.fuse_groups_autograd_hip(x, groups)
        return [self.fuse(self._views(x, g)) for g in groups]

    @staticmethod
    def _views(x, g):
        """x[:, g] for a list of view indices - as a slice when they form a run (no index tensor: capturable into a hipGraph)."""
        g = list(g)
        return x[:, g[0]:g[0] + len(g)] if g == list(range(g[0], g[0] + len(g))) else x[:, g]

    @staticmethod
    def _bn2d_rows(bn, rows, relu=True):
        return bn_act_rows(bn, rows, 0.0 if relu else 1.0)

    def _trunk_autograd_hip(self, img):
        """ResNet-50 trunk with an autograd graph on the HIP kernels (resnet_rows_autograd)."""
        fe = self.feature_extraction
        return resnet_rows_autograd(fe[0], fe[1], fe[3], [fe[4], fe[5], fe[6], fe[7]], img, slope=0.0)[-1]

    def _head_autograd_hip(self, head, z):
        """A head (nn.Sequential of ConvTranspose3d / Conv3d / BatchNorm3d / LeakyReLU / ReLU, models/encoder.py:16-34) with an
        autograd graph: every convolution forward, data gradient and weight gradient on the HIP GEMM / wgrad kernels, the
        normalisations and activations as torch ops on the same channels-last rows."""
        rows = self._rows(z)
        mods = list(head)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.ConvTranspose3d):
                rows = co.convT3d_k4s2p1_rows(rows, m.weight, m.bias)
            elif isinstance(m, nn.Conv3d):
                rows = co.conv3x3x3_rows_any(rows, m.weight, m.bias)
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                nxt = mods[i + 1] if i + 1 < len(mods) else None      # BatchNorm + the activation behind it: one fused pass each way (csrc/bnorm.hip)
                if isinstance(nxt, nn.LeakyReLU):
                    rows, i = bn_act_rows(m, rows, nxt.negative_slope), i + 1
                elif isinstance(nxt, nn.ReLU):
                    rows, i = bn_act_rows(m, rows, 0.0), i + 1
                else:
                    rows = bn_act_rows(m, rows)
            elif isinstance(m, nn.LeakyReLU):
                rows = torch.nn.functional.leaky_relu(rows, m.negative_slope)
            elif isinstance(m, nn.ReLU):
                rows = torch.relu(rows)
            else:
                raise TypeError("unexpected layer in head: %r" % (m,))
            i += 1
        return rows.permute(0, 4, 1, 2, 3)

    # ---------------------------------------------------------------- fused HIP inference path
    @staticmethod
    def _rows(x):
        """[N,C,D,H,W] -> channels-last rows tensor [N,D,H,W,C] (no copy when already channels_last_3d)."""
        r = x.permute(0, 2, 3, 4, 1)
        return r if r.is_contiguous() else r.contiguous()

    LIFT_Z, LIFT_C = 32, 64          # z_2d.view(-1, 64, 32, H, W): trunk channel c = c3d*32 + z (models/encoder.py:49)
    TRUNK_WINO_MIN_PLANES = 256      # bottleneck width from which the stride-1 3x3 convolutions run as Winograd (sweep: profiles/TUNING_LOG.md)

    def _trunk_packed(self):
        """Packed weights of ResNet layers 1-4 for the GEMM kernel. The 2048-wide residual stream of layer4 is kept in
        the permuted channel order j = z*64 + c3d (original channel c3d*32 + z): a pure relabelling (applied to Cout of
        layer4's conv3/downsample and to Cin of the following conv1) that lets the last GEMM store the lifted
        [N,32,H,W,64] volume with contiguous 256-byte segments."""
        fe = self.feature_extraction
        src = [t for li in (4, 5, 6, 7) for blk in fe[li] for t in list(blk.parameters()) + list(blk.buffers())]

        def build():
            dev = fe[0].weight.device
            j = torch.arange(self.LIFT_Z * self.LIFT_C, device=dev)
            perm = (j % self.LIFT_C) * self.LIFT_Z + j // self.LIFT_C
            blocks = []
            for li in (4, 5, 6, 7):
                for bi, blk in enumerate(fe[li]):
                    pin = perm if (li == 7 and bi > 0) else None          # block input already in permuted order
                    pout = perm if li == 7 else None
                    w1 = blk.conv1.weight.detach()[:, :, 0, 0]
                    if pin is not None:
                        w1 = w1[:, pin]
                    w2, taps2 = co.pack_conv2d_weight(blk.conv2.weight)
                    w3 = blk.conv3.weight.detach()[:, :, 0, 0]
                    a3 = co.bn_affine(blk.bn3)
                    if pout is not None:
                        w3, a3 = w3[pout], (a3[0][pout].contiguous(), a3[1][pout].contiguous())
                    # stride-1 3x3 convolutions with GEMM-sized channel counts (layer3 / layer4: K = planes >= 256 per Winograd point) run as
                    # Winograd F(2x2, 3x3), csrc/winograd.hip with one "depth" tap (layer1/2 are too narrow: measured)
                    u2 = co.wino_pack_packed(w2) if (blk.conv2.stride[0] == 1 and w1.shape[0] >= self.TRUNK_WINO_MIN_PLANES) else None
                    d = {"w1": w1[None].contiguous(), "a1": co.bn_affine(blk.bn1), "w2": w2, "u2": u2, "taps2": taps2, "a2": co.bn_affine(blk.bn2),
                         "stride": blk.conv2.stride[0], "w3": w3[None].contiguous(), "a3": a3, "planes": w1.shape[0], "ds": None}
                    if blk.downsample is not None:
                        wd = blk.downsample[0].weight.detach()[:, :, 0, 0]
                        ad = co.bn_affine(blk.downsample[1])
                        if pout is not None:
                            wd, ad = wd[pout], (ad[0][pout].contiguous(), ad[1][pout].contiguous())
                        d["ds"] = (wd[None].contiguous(), ad, blk.downsample[0].stride[0])
                    blocks.append(d)
            return blocks
        return self._trunk_cache.get(src, build)

    @staticmethod
    def _trunk_out_hw(H, W):
        f = lambda v, k, s, p: (v + 2 * p - k) // s + 1
        return tuple(f(f(f(v, 7, 2, 3), 3, 2, 1), 3, 2, 1) for v in (H, W))        # stem conv, max-pool, layer2's stride-2 3x3

    @_lib.on_tensor_device
    def _trunk_hip(self, img):
        """The whole ResNet-50 trunk on the fp32 matrix cores: stem (patch gather + GEMM, max-pool kernel), layers 1-4 as
        im2col-free implicit GEMMs (1x1 = plain GEMM, 3x3 = 9 taps, strides via the input-stride argument), BN folded, ReLU
        and the residual add in the epilogue, activations NHWC, the 2D->3D lift fused into the last store.
        img [N,3,H,W] -> lifted volume rows [N,32,H/8,W/8,64] (input of conv1).
        (Round 2 A/B: running the views of one scene as concurrent image groups on side streams / parallel hipGraph branches made the
        step SLOWER - 12.09 -> 12.32 ms with 2 groups, 14.37 ms with 5: the branches do not overlap on this runtime and every extra
        kernel costs its ~8 us; DESIGN.md tuning log.)"""
        N, _, H, W = img.shape
        Ho, Wo = self._trunk_out_hw(H, W)
        out = torch.empty(N, self.LIFT_Z, Ho, Wo, self.LIFT_C, dtype=torch.float32, device=img.device)
        self._trunk_hip_group(img, out)
        return out

    def _stem_packed(self):
        conv0, bn0 = self.feature_extraction[0], self.feature_extraction[1]
        kh, kw = conv0.kernel_size
        Kp = ((kh * kw * conv0.in_channels + 31) // 32) * 32

        def build_stem():
            w = conv0.weight.detach().permute(0, 2, 3, 1).reshape(conv0.out_channels, -1)          # [Cout][(ky,kx,c)]
            return (co.pad_cin(w[None].contiguous(), Kp),) + co.bn_affine(bn0)
        return self._stem_cache.get([conv0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var], build_stem)

    def _trunk_hip_group(self, img, dst):
        """One group of images through the trunk (see _trunk_hip); the last GEMM stores the lifted volume into dst [n,32,H/8,W/8,64]."""
        fe = self.feature_extraction
        dev = img.device
        T1 = [(0, 0, 0)]
        # ---- stem: 7x7/s2 conv as patch-gather + one-tap GEMM (BN + ReLU folded), then the 3x3/s2 max-pool, all channels-last
        conv0, bn0, pool = fe[0], fe[1], fe[3]
        kh, kw = conv0.kernel_size
        Kp = ((kh * kw * conv0.in_channels + 31) // 32) * 32
        w0, sc0, sh0 = self._stem_packed()
        N, Ci, Hi, Wi = img.shape
        s0, p0 = conv0.stride[0], conv0.padding[0]
        Hc, Wc = (Hi + 2 * p0 - kh) // s0 + 1, (Wi + 2 * p0 - kw) // s0 + 1
        img_c = img.contiguous()
        patches = torch.empty(N * Hc * Wc, Kp, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().forge_im2col_nchw(_lib.ptr(img_c), _lib.ptr(patches), N, Ci, Hi, Wi, kh, kw, s0, p0, Kp, _lib.current_stream()),
                   "forge_im2col_nchw")
        c0 = torch.empty(N, Hc, Wc, conv0.out_channels, dtype=torch.float32, device=dev)
        co.conv_igemm(patches, Kp, Kp, None, 0, 0, w0, None, sc0, sh0, 0.0, None, None, None, c0, None,
                      (N, 1, Hc, Wc), (1, Hc, Wc), conv0.out_channels, conv0.out_channels, T1, epilogue=co.EPI_AFFINE_ACT)
        pk, ps, pp = pool.kernel_size, pool.stride, pool.padding
        H, W = (Hc + 2 * pp - pk) // ps + 1, (Wc + 2 * pp - pk) // ps + 1
        xr = torch.empty(N, H, W, conv0.out_channels, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().forge_maxpool2d_nhwc(_lib.ptr(c0), _lib.ptr(xr), N, Hc, Wc, conv0.out_channels, pk, ps, pp, _lib.current_stream()),
                   "forge_maxpool2d_nhwc")
        blocks = self._trunk_packed()
        for bi, b in enumerate(blocks):
            last = bi == len(blocks) - 1
            Cin, P, s = xr.shape[-1], b["planes"], b["stride"]
            Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
            y1 = torch.empty(N, H, W, P, dtype=torch.float32, device=dev)
            co.conv_igemm(xr, Cin, Cin, None, 0, 0, b["w1"], None, b["a1"][0], b["a1"][1], 0.0, None, None, None, y1, None,
                          (N, 1, H, W), (1, H, W), P, P, T1, epilogue=co.EPI_AFFINE_ACT)
            y2 = torch.empty(N, Ho, Wo, P, dtype=torch.float32, device=dev)
            if b["u2"] is not None and co.wino_enabled() and H % 2 == 0 and W % 2 == 0:
                V = co.wino_input(y1, P, P, N, 1, H, W)
                Mm = torch.empty(16, N * (H // 2) * (W // 2), P, dtype=torch.float32, device=dev)
                co.wino_gemm(V, P, None, 0, b["u2"], Mm, N, 1, H // 2, W // 2, P)
                co.wino_output(Mm, None, b["a2"][0], b["a2"][1], 0.0, None, None, None, y2, None, None, N, 1, H, W, P, P, co.EPI_AFFINE_ACT)
            else:
                co.conv_igemm(y1, P, P, None, 0, 0, b["w2"], None, b["a2"][0], b["a2"][1], 0.0, None, None, None, y2, None,
                              (N, 1, Ho, Wo), (1, H, W), P, P, b["taps2"], istride=s, epilogue=co.EPI_AFFINE_ACT)
            if b["ds"] is not None:
                wd, ad, sd = b["ds"]
                idn = torch.empty(N, Ho, Wo, 4 * P, dtype=torch.float32, device=dev)
                co.conv_igemm(xr, Cin, Cin, None, 0, 0, wd, None, ad[0], ad[1], 1.0, None, None, None, idn, None,
                              (N, 1, Ho, Wo), (1, H, W), 4 * P, 4 * P, T1, istride=sd, epilogue=co.EPI_AFFINE_ACT)
            else:
                idn = xr
            if last:
                assert dst.shape == (N, self.LIFT_Z, Ho, Wo, self.LIFT_C) and dst.is_contiguous()
                out = dst
            else:
                out = torch.empty(N, Ho, Wo, 4 * P, dtype=torch.float32, device=dev)
            co.conv_igemm(y2, P, P, None, 0, 0, b["w3"], None, b["a3"][0], b["a3"][1], 0.0, idn, None, None, out, None,
                          (N, 1, Ho, Wo), (1, Ho, Wo), 4 * P, 4 * P, T1, epilogue=co.EPI_AFFINE_ACT, lift=self.LIFT_Z if last else 0)
            xr, H, W = out, Ho, Wo
        return xr

    def _conv1_hip(self, vol_rows):
        """conv1 = Conv3d(64,128,3,p1)+BN+LeakyReLU as one GEMM (models/encoder.py:36-40). vol_rows [N,D,H,W,64]."""
        conv, bn = self.conv1[0], self.conv1[1]
        w, bias, sc, sh, U = self._c1_cache.get(
            [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var],
            lambda: (co.pack_conv3d_weight(conv.weight), conv.bias.detach().contiguous()) + co.bn_affine(bn) + (co.wino_pack_weight(conv.weight),))
        n, D, H, W, C = vol_rows.shape
        out = torch.empty(n, D, H, W, 128, dtype=torch.float32, device=vol_rows.device)
        if co.wino_applies(co.TAPS_3x3x3, 1, n, D, H, W, C, 0, 128):       # Winograd F(2x2,3x3) x 3 depth taps, BN + LeakyReLU in the inverse transform
            V = co.wino_input(vol_rows, C, C, n, D, H, W)
            R = n * D * (H // 2) * (W // 2)
            hf = co.wino_half_applies(R, 128, C)                        # row stage of the inverse transform in the GEMM epilogue: 8 planes instead of 16
            Mm = torch.empty(8 if hf else 16, R, 128, dtype=torch.float32, device=vol_rows.device)
            co.wino_gemm(V, C, None, 0, U, Mm, n, D, H // 2, W // 2, 128)
            co.wino_output(Mm, bias, sc, sh, 0.01, None, None, None, out, None, None, n, D, H, W, 128, 128, co.EPI_AFFINE_ACT)
        else:
            co.conv_igemm(vol_rows, C, C, None, 0, 0, w, bias, sc, sh, 0.01, None, None, None, out, None,
                          (n, D, H, W), (D, H, W), 128, 128, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
        return out.permute(0, 4, 1, 2, 3)

    def _heads_packed_T(self):
        """Transposed / padded packed head weights and BatchNorm scales for _HeadsFrozen.backward (cached with the forward pack)."""
        self._heads_hip_pack()
        p = self._heads_cache.val
        if "ct_wT" not in p:
            fh, dh = self.features_head, self.density_head
            wct = torch.cat([fh[0].weight, dh[0].weight], dim=1).detach()                      # [128, 64, 4, 4, 4]
            tr16 = lambda w: co.pad_last(w.transpose(1, 2).contiguous(), 16)                   # [27][Cin][Cout -> 16]
            p.update({"ct_wT": wct.reshape(wct.shape[0], 64, 64).permute(2, 0, 1).contiguous(),        # [k][Cin=128][Cout=64]
                      "ct_taps": [(kz - 1, ky - 1, kx - 1) for kz in range(4) for ky in range(4) for kx in range(4)],
                      "ct_scale": p["ct_aff"][0], "f4_scale": p["f4"][0], "d4_scale": p["d4"][0],
                      "f3_wT": tr16(p["f3_w"]), "d3_wT": tr16(p["d3_w"])})
        return p

    def _heads_hip_pack(self):
        fh, dh = self.features_head, self.density_head
        src = [fh[0].weight, fh[0].bias, dh[0].weight, dh[0].bias, fh[3].weight, fh[3].bias, dh[3].weight, dh[3].bias,
               dh[6].weight, dh[6].bias] + [t for bn in (fh[1], dh[1], fh[4], dh[4]) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]

        def build():
            ct = co.convT_phases_merged(torch.cat([fh[0].weight, dh[0].weight], dim=1), 1, 3)  # Cout = 32 + 32; 8 phases x 8 taps, one launch
            s1 = [torch.cat(v) for v in zip(co.bn_affine(fh[1]), co.bn_affine(dh[1]))]
            w6 = co.pack_conv3d_weight(dh[6].weight)                                       # [27][1][8]: direct (vector-ALU) kernel
            return {"ct": ct, "ct_b": torch.cat([fh[0].bias, dh[0].bias]).detach().contiguous(), "ct_aff": s1,
                    "ct_f": co.convT_phases_merged(fh[0].weight, 1, 3), "ct_d": co.convT_phases_merged(dh[0].weight, 1, 3),
                    "f3_w": co.pack_conv3d_weight(fh[3].weight), "f3_b": fh[3].bias.detach().contiguous(), "f4": co.bn_affine(fh[4]),
                    "d3_w": co.pack_conv3d_weight(dh[3].weight), "d3_b": dh[3].bias.detach().contiguous(), "d4": co.bn_affine(dh[4]),
                    "d6_w": w6, "d6_b": dh[6].bias.detach().contiguous()}
        return self._heads_cache.get(src, build)

    def _heads_hip(self, z, which="both", keep=False):
        """The heads (models/encoder.py:16-34) on a fused volume, returns (features | None, density | None).
        which = "both": the two ConvTranspose3d(128,32,4,s2,p1)+BN+LReLU run as ONE N=64 launch covering the 8 output phases x 8
        taps, then Conv3d(32,16)+BN and Conv3d(32,8)+BN+LReLU read their 32-channel halves of the shared [..,64] tensor in place, then
        Conv3d(8,1)+ReLU. which = "features" / "density": that head alone (N=32 transposed convolution), as the reference computes it.
        keep: also return the intermediate activations (up, d8) for the hand-written backward of _HeadsFrozen."""
        p = self._heads_hip_pack()
        n, C, D, H, W = z.shape
        D2, H2, W2 = 2 * D, 2 * H, 2 * W
        dev = z.device
        xr = self._rows(z)
        if which == "both":
            taps, wct, bct, sc, sh, Nup, fo, do = p["ct"][0], p["ct"][1], p["ct_b"], p["ct_aff"][0], p["ct_aff"][1], 64, 0, 32
        elif which == "features":
            taps, wct, bct, sc, sh, Nup, fo, do = p["ct_f"][0], p["ct_f"][1], p["ct_b"][:32], p["ct_aff"][0][:32], p["ct_aff"][1][:32], 32, 0, None
        elif which == "density":
            taps, wct, bct, sc, sh, Nup, fo, do = p["ct_d"][0], p["ct_d"][1], p["ct_b"][32:], p["ct_aff"][0][32:], p["ct_aff"][1][32:], 32, None, 0
        else:
            raise ValueError("which must be 'both', 'features' or 'density'")
        up = torch.empty(n, D2, H2, W2, Nup, dtype=torch.float32, device=dev)
        co.conv_igemm(xr, C, C, None, 0, 0, wct, bct, sc, sh, 0.01, None, None, None, up, None,
                      (n, D, H, W), (D, H, W), Nup, Nup, taps, out_grid=(D2, H2, W2), ostride=2, phase=(-1, -1, -1),
                      epilogue=co.EPI_AFFINE_ACT)
        g2, ig2 = (n, D2, H2, W2), (D2, H2, W2)
        feat = dens = None
        if fo is not None:
            feat = torch.empty(n, D2, H2, W2, 16, dtype=torch.float32, device=dev)
            co.conv_igemm(up[..., fo:], 32, Nup, None, 0, 0, p["f3_w"], p["f3_b"], p["f4"][0], p["f4"][1], 1.0, None, None, None, feat, None,
                          g2, ig2, 16, 16, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
            feat = feat.permute(0, 4, 1, 2, 3)
        if do is not None:
            d8 = torch.empty(n, D2, H2, W2, 8, dtype=torch.float32, device=dev)
            co.conv_igemm(up[..., do:], 32, Nup, None, 0, 0, p["d3_w"], p["d3_b"], p["d4"][0], p["d4"][1], 0.01, None, None, None, d8, None,
                          g2, ig2, 8, 8, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
            dens = torch.empty(n, D2, H2, W2, 1, dtype=torch.float32, device=dev)
            co.conv_direct(d8, 8, p["d6_w"], p["d6_b"], 0.0, dens, g2, 8, 1, co.TAPS_3x3x3)   # Conv3d(8, 1) + ReLU: 216 MACs per voxel
            dens = dens.permute(0, 4, 1, 2, 3)
        if keep:
            return feat, dens, up, d8
        return feat, dens

    def forward(self, x):
        raise NotImplementedError
