"""a1/a5 — 2D->3D feature lift, fusion wrapper and the density / render-feature heads.

Mirror of the reference's models/encoder.py (Encoder3D :8-68, get_resnet50 :71-78): same attribute
names (`feature_extraction`, `features_head`, `density_head`, `conv1`, `fusion_feature`), same
methods (`get_feat3D`, `get_density3D`, `get_render_features`, `fuse`) and the same state_dict keys
(SURVEY.md Appendix B). torchvision is not a dependency: the ResNet-50 trunk is built here with
torchvision's module names so published checkpoints load with strict=True.
"""
import weakref

import torch
import torch.nn as nn

from . import convops as co
from .fusion import ConvGRU_3D, hip_inference


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


class Encoder3D(nn.Module):
    """models/encoder.py:8-68."""

    def __init__(self, config):
        super().__init__()
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
        z_2d = self.feature_extraction(img)
        B, C, H, W = z_2d.shape
        z_3d = z_2d.view(-1, 64, 32, H, W)
        if hip_inference(self, z_3d):
            return self._conv1_hip(z_3d)
        return self.conv1(z_3d)

    def get_density3D(self, z_3d):
        if hip_inference(self, z_3d):
            return self._heads_hip(z_3d)[1]
        return self.density_head(z_3d)

    def get_render_features(self, x):
        if hip_inference(self, x):
            return self._heads_hip(x)[0]
        return self.features_head(x)

    def fuse(self, x):
        """x [b,t,c,d,h,w] -> [b,c,d,h,w] (models/encoder.py:59-63)"""
        if hip_inference(self, x):
            return self.fusion_feature.fuse_hip(x)
        return self.fusion_feature(x, [self.fusion_feature.fusion_conv(x.mean(dim=1))])

    # ---------------------------------------------------------------- fused HIP inference path
    @staticmethod
    def _rows(x):
        """[N,C,D,H,W] -> channels-last rows tensor [N,D,H,W,C] (no copy when already channels_last_3d)."""
        r = x.permute(0, 2, 3, 4, 1)
        return r if r.is_contiguous() else r.contiguous()

    def _conv1_hip(self, z_3d):
        """conv1 = Conv3d(64,128,3,p1)+BN+LeakyReLU as one GEMM (models/encoder.py:36-40)."""
        conv, bn = self.conv1[0], self.conv1[1]
        if not hasattr(self, "_c1_cache"):
            self._c1_cache = co.PackCache()
        w, bias, sc, sh = self._c1_cache.get(
            [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var],
            lambda: (co.pack_conv3d_weight(conv.weight), conv.bias.detach().contiguous()) + co.bn_affine(bn))
        n, C, D, H, W = z_3d.shape
        xr = self._rows(z_3d)
        out = torch.empty(n, D, H, W, 128, dtype=torch.float32, device=z_3d.device)
        co.conv_igemm(xr, C, C, None, 0, 0, w, bias, sc, sh, 0.01, None, None, None, out, None,
                      (n, D, H, W), (D, H, W), 128, 128, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
        return out.permute(0, 4, 1, 2, 3)

    def _heads_hip(self, z):
        """Both heads (models/encoder.py:16-34) on one fused volume: the two ConvTranspose3d(128,32,4,s2,p1)+BN+LReLU
        run as ONE N=64 GEMM per output phase (8 phases x 8 taps), then Conv3d(32,16)+BN and
        Conv3d(32,8)+BN+LReLU read their 32-channel halves of the shared [..,64] tensor in place, then Conv3d(8,1)+ReLU.
        Results are cached per input tensor so get_density3D / get_render_features share the work."""
        memo = getattr(self, "_heads_memo", None)
        if memo is not None and memo[0]() is z and memo[1] == z._version:
            return memo[2]
        fh, dh = self.features_head, self.density_head
        if not hasattr(self, "_heads_cache"):
            self._heads_cache = co.PackCache()
        src = [fh[0].weight, fh[0].bias, dh[0].weight, dh[0].bias, fh[3].weight, fh[3].bias, dh[3].weight, dh[3].bias,
               dh[6].weight, dh[6].bias] + [t for bn in (fh[1], dh[1], fh[4], dh[4]) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]

        def build():
            ct = co.convT3d_k4s2p1_phases(torch.cat([fh[0].weight, dh[0].weight], dim=1))      # Cout = 32 + 32
            s1 = [torch.cat(v) for v in zip(co.bn_affine(fh[1]), co.bn_affine(dh[1]))]
            w6 = co.pad_cin(co.pack_conv3d_weight(dh[6].weight), 16)
            return {"ct": ct, "ct_b": torch.cat([fh[0].bias, dh[0].bias]).detach().contiguous(), "ct_aff": s1,
                    "f3_w": co.pack_conv3d_weight(fh[3].weight), "f3_b": fh[3].bias.detach().contiguous(), "f4": co.bn_affine(fh[4]),
                    "d3_w": co.pack_conv3d_weight(dh[3].weight), "d3_b": dh[3].bias.detach().contiguous(), "d4": co.bn_affine(dh[4]),
                    "d6_w": w6, "d6_b": dh[6].bias.detach().contiguous(),
                    "one": torch.ones(1, device=z.device), "zero": torch.zeros(1, device=z.device)}
        p = self._heads_cache.get(src, build)
        n, C, D, H, W = z.shape
        D2, H2, W2 = 2 * D, 2 * H, 2 * W
        dev = z.device
        xr = self._rows(z)
        up = torch.empty(n, D2, H2, W2, 64, dtype=torch.float32, device=dev)
        for (pz, py, px), taps, wp in p["ct"]:
            co.conv_igemm(xr, C, C, None, 0, 0, wp, p["ct_b"], p["ct_aff"][0], p["ct_aff"][1], 0.01, None, None, None, up, None,
                          (n, D, H, W), (D, H, W), 64, 64, taps, out_grid=(D2, H2, W2), ostride=2, phase=(pz, py, px),
                          epilogue=co.EPI_AFFINE_ACT)
        g2, ig2 = (n, D2, H2, W2), (D2, H2, W2)
        feat = torch.empty(n, D2, H2, W2, 16, dtype=torch.float32, device=dev)
        co.conv_igemm(up, 32, 64, None, 0, 0, p["f3_w"], p["f3_b"], p["f4"][0], p["f4"][1], 1.0, None, None, None, feat, None,
                      g2, ig2, 16, 16, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
        d8 = torch.zeros(n, D2, H2, W2, 16, dtype=torch.float32, device=dev)          # 8 real channels, zero-padded to the 16-wide K-step
        co.conv_igemm(up[..., 32:], 32, 64, None, 0, 0, p["d3_w"], p["d3_b"], p["d4"][0], p["d4"][1], 0.01, None, None, None, d8, None,
                      g2, ig2, 8, 16, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
        dens = torch.empty(n, D2, H2, W2, 1, dtype=torch.float32, device=dev)
        co.conv_igemm(d8, 16, 16, None, 0, 0, p["d6_w"], p["d6_b"], p["one"], p["zero"], 0.0, None, None, None, dens, None,
                      g2, ig2, 1, 1, co.TAPS_3x3x3, epilogue=co.EPI_AFFINE_ACT)
        res = (feat.permute(0, 4, 1, 2, 3), dens.permute(0, 4, 1, 2, 3))
        self._heads_memo = (weakref.ref(z), z._version, res)
        return res

    def forward(self, x):
        raise NotImplementedError
