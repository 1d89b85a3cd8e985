"""a1/a5 — 2D->3D feature lift, fusion wrapper and the density / render-feature heads.

Mirror of the reference's models/encoder.py (Encoder3D :8-68, get_resnet50 :71-78): same attribute
names (`feature_extraction`, `features_head`, `density_head`, `conv1`, `fusion_feature`), same
methods (`get_feat3D`, `get_density3D`, `get_render_features`, `fuse`) and the same state_dict keys
(SURVEY.md Appendix B). torchvision is not a dependency: the ResNet-50 trunk is built here with
torchvision's module names so published checkpoints load with strict=True.
"""
import torch.nn as nn

from .fusion import ConvGRU_3D


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
        return self.conv1(z_3d)

    def get_density3D(self, z_3d):
        return self.density_head(z_3d)

    def get_render_features(self, x):
        return self.features_head(x)

    def fuse(self, x):
        """x [b,t,c,d,h,w] -> [b,c,d,h,w] (models/encoder.py:59-63)"""
        return self.fusion_feature(x, [self.fusion_feature.fusion_conv(x.mean(dim=1))])

    def forward(self, x):
        raise NotImplementedError
