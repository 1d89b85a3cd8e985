"""2-D pose estimator (ResNet-50 FPN + cross/self-attention): the reference's attribute surface (`encoder_traj_2d`) and state_dict keys
(`encoder_traj_2d.*`, 40.19 M parameters); architecture re-expressed from models/pose_estimator_2d.py:10-275 and the Perceiver-style attention
blocks of models/model_utils.py:258-427 (einops is not required).

On the MI355X every convolution of the module - the FPN's LeakyReLU ResNet-50, its lateral / top / smoothing layers and the four stride-2
convolutions of `conv` - runs on libforge_hip.so with HIP BatchNorm (round 5; FPN.forward_rows, PoseEstimator2D._conv_rows; in eval mode without
an autograd graph as the inference schedule of forge_amd/frozen.py: one launch per convolution, BatchNorm folded); the six attention
blocks stay stock torch (rocBLAS GEMMs, softmax, LayerNorm). One path: host tensors raise (the stock-torch evaluation of the same modules that
tests and probes compare against lives in tools/stock_pose.py)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .convops import PackCache as _PackCache, PackedModule as _PackedModule


def sincos_pos_embed_2d(embed_dim, grid_size):
    """models/model_utils.py:9-56 (MAE-style, float64): first half of the channels encodes the column
    index, second half the row index; each half is [sin | cos] over embed_dim/4 frequencies."""
    def one_axis(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = pos.reshape(-1)[:, None] * omega[None]
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    cols, rows = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32))
    return np.concatenate([one_axis(embed_dim // 2, cols), one_axis(embed_dim // 2, rows)], axis=1)


class MultiHeadAttention(nn.Module):
    """models/model_utils.py:258-342 (scaled dot-product, separate q/k/v/o projections)."""

    def __init__(self, num_heads, num_q_input_channels, num_kv_input_channels, num_qk_channels=None,
                 num_v_channels=None, num_output_channels=None, dropout=0.0):
        super().__init__()
        num_qk_channels = num_qk_channels or num_q_input_channels
        num_v_channels = num_v_channels or num_qk_channels
        num_output_channels = num_output_channels or num_q_input_channels
        if num_qk_channels % num_heads or num_v_channels % num_heads:
            raise ValueError("channel counts must be divisible by num_heads")
        self.dp_scale = (num_qk_channels // num_heads) ** -0.5
        self.num_heads = num_heads
        self.q_proj = nn.Linear(num_q_input_channels, num_qk_channels)
        self.k_proj = nn.Linear(num_kv_input_channels, num_qk_channels)
        self.v_proj = nn.Linear(num_kv_input_channels, num_v_channels)
        self.o_proj = nn.Linear(num_v_channels, num_output_channels)
        self.dropout = nn.Dropout(dropout)

    def _heads(self, x):
        b, n, c = x.shape
        return x.reshape(b, n, self.num_heads, c // self.num_heads).permute(0, 2, 1, 3).reshape(b * self.num_heads, n, -1)

    def forward(self, x_q, x_k, x_v, pad_mask=None, attn_mask=None):
        if attn_mask is not None:
            raise NotImplementedError("attention masks not supported")
        b = x_q.shape[0]
        q, k, v = self._heads(self.q_proj(x_q)), self._heads(self.k_proj(x_k)), self._heads(self.v_proj(x_v))
        attn = torch.bmm(q, k.transpose(1, 2)) * self.dp_scale
        if pad_mask is not None:
            mask = pad_mask[:, None, None, :].expand(b, self.num_heads, 1, pad_mask.shape[-1]).reshape(b * self.num_heads, 1, -1)
            attn = attn.masked_fill(mask, -torch.finfo(attn.dtype).max)
        attn = self.dropout(attn.softmax(dim=-1))
        o = torch.bmm(attn, v)
        o = o.reshape(b, self.num_heads, o.shape[1], -1).permute(0, 2, 1, 3).reshape(b, o.shape[1], -1)
        return self.o_proj(o)


class MLP_attention(nn.Module):
    """models/model_utils.py:345-356"""

    def __init__(self, num_channels, widening_factor):
        super().__init__()
        self.mlp = nn.Sequential(nn.LayerNorm(num_channels), nn.Linear(num_channels, widening_factor * num_channels),
                                 nn.GELU(), nn.Linear(widening_factor * num_channels, num_channels))

    def forward(self, x):
        return self.mlp(x)


class CrossAttention(nn.Module):
    """models/model_utils.py:359-396 — note: the MLP output REPLACES x (no residual around the MLP)."""

    def __init__(self, num_heads, num_q_input_channels, num_kv_input_channels, mlp_ratio=1,
                 num_qk_channels=None, num_v_channels=None, dropout=0.0):
        super().__init__()
        self.q_norm = nn.LayerNorm(num_q_input_channels)
        self.k_norm = nn.LayerNorm(num_kv_input_channels)
        self.v_norm = nn.LayerNorm(num_kv_input_channels)
        self.attention = MultiHeadAttention(num_heads, num_q_input_channels, num_kv_input_channels,
                                            num_qk_channels, num_v_channels, dropout=dropout)
        self.mlp = MLP_attention(num_q_input_channels, mlp_ratio)

    def forward(self, x_q, x_k, x_v, pad_mask=None, attn_mask=None, residual=False):
        x_q, x_k, x_v = self.q_norm(x_q), self.k_norm(x_k), self.v_norm(x_v)
        a = self.attention(x_q, x_k, x_v, pad_mask=pad_mask, attn_mask=attn_mask)
        if torch.is_tensor(residual):
            a = residual + a
        elif residual is True:
            a = x_q + a
        return self.mlp(a)


class SelfAttention(nn.Module):
    """models/model_utils.py:399-427"""

    def __init__(self, num_heads, num_channels, mlp_ratio=1, num_qk_channels=None, num_v_channels=None, dropout=0.0):
        super().__init__()
        self.norm = nn.LayerNorm(num_channels)
        self.attention = MultiHeadAttention(num_heads, num_channels, num_channels, num_qk_channels, num_v_channels,
                                            dropout=dropout)
        self.mlp = MLP_attention(num_channels, mlp_ratio)

    def forward(self, x, pad_mask=None, attn_mask=None):
        x = self.norm(x)
        return self.mlp(x + self.attention(x, x, x, pad_mask=pad_mask, attn_mask=attn_mask))


class _BottleneckLReLU(nn.Module):
    """models/pose_estimator_2d.py:237-275 — ResNet bottleneck with LeakyReLU activations."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.leakyrelu = nn.LeakyReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.leakyrelu(self.bn1(self.conv1(x)))
        out = self.leakyrelu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.leakyrelu(out + res)


def _res_layer(inplanes, planes, blocks, stride):
    ds = None
    if stride != 1 or inplanes != planes * 4:
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    return nn.Sequential(_BottleneckLReLU(inplanes, planes, stride, ds),
                         *[_BottleneckLReLU(planes * 4, planes) for _ in range(1, blocks)])


class FPN(_PackedModule):
    """models/pose_estimator_2d.py:91-136: ResNet-50 (LeakyReLU) bottom-up, only the stride-16 level
    p4 = smooth1(upsample(toplayer(c5)) + latlayer1(c4)) is used; smooth2/3, latlayer2/3 exist unused.
    The ImageNet weights the reference downloads arrive via load_state_dict."""

    def __init__(self):
        super().__init__()
        self._res_cache, self._head_cache = _PackCache(), _PackCache()      # inference launch arguments (forge_amd/frozen.py)
        self.toplayer = nn.Conv2d(2048, 256, 1)
        self.layer0 = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
                                    nn.LeakyReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1))
        self.layer1 = nn.Sequential(_res_layer(64, 64, 3, 1))
        self.layer2 = nn.Sequential(_res_layer(256, 128, 4, 2))
        self.layer3 = nn.Sequential(_res_layer(512, 256, 6, 2))
        self.layer4 = nn.Sequential(_res_layer(1024, 512, 3, 2))
        self.smooth1 = nn.Conv2d(256, 256, 3, padding=1)
        self.smooth2 = nn.Conv2d(256, 256, 3, padding=1)
        self.smooth3 = nn.Conv2d(256, 256, 3, padding=1)
        self.latlayer1 = nn.Conv2d(1024, 256, 1)
        self.latlayer2 = nn.Conv2d(512, 256, 1)
        self.latlayer3 = nn.Conv2d(256, 256, 1)
        for blk in (self.layer0, self.layer1, self.layer2, self.layer3, self.layer4):
            for m in blk.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu")

    def forward(self, x):
        """[n,3,H,W] -> p4 [n,256,H/16,W/16] (an NCHW view of forward_rows' NHWC rows)."""
        return self.forward_rows(x).permute(0, 3, 1, 2)

    def forward_rows(self, x):
        """The same pyramid level as NHWC rows [n,h,w,256] on libforge_hip.so: the LeakyReLU ResNet-50 through encoder.resnet_rows_autograd (MFMA
        implicit-GEMM convolutions forward / data / weight gradient, HIP BatchNorm), the lateral / top / smoothing convolutions through
        convops.conv2d_rows; only the 8x8 -> 16x16 bilinear up-sampling and the max-pool are torch ops. (On MIOpen these convolutions took whatever
        solver its find step landed on in that process - asm Winograd in one run, `naive_conv_*` in the next.)"""
        from . import convops as co
        from . import frozen as fz
        from .encoder import resnet_rows_autograd
        from .fusion import require_hip_input
        require_hip_input("PoseEstimator2D's FPN", x)
        l0 = self.layer0
        if fz.frozen_ok(x, self):
            # inference: every convolution ONE launch with bias / folded BatchNorm / residual / LeakyReLU in its epilogue; the lateral convolution
            # takes the up-sampled top level as its residual (p4 = upsample(toplayer(c5)) + latlayer1(c4))
            stages = [self.layer1[0], self.layer2[0], self.layer3[0], self.layer4[0]]
            P = fz.pack_resnet(self._res_cache, l0[0], l0[1], stages, l0[2].negative_slope)
            _, _, c4, c5 = fz.run_resnet(P, l0[0], l0[3], x, l0[2].negative_slope)                 # rows [N,1,h,w,C]
            heads = (self.latlayer1, self.toplayer, self.smooth1)
            lat, top, smooth = self._head_cache.get([t for m in heads for t in (m.weight, m.bias)],
                                                    lambda: [fz.pack_layer(m, None, 1.0, force_affine=(m is self.latlayer1)) for m in heads])
            t = fz.run_layer(top, c5)[:, 0]
            up = F.interpolate(t.permute(0, 3, 1, 2), size=c4.shape[2:4], mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
            return fz.run_layer(smooth, fz.run_layer(lat, c4, residual=up.contiguous()[:, None]))[:, 0]
        _, _, c4, c5 = resnet_rows_autograd(l0[0], l0[1], l0[3], [self.layer1[0], self.layer2[0], self.layer3[0], self.layer4[0]], x,
                                            slope=l0[2].negative_slope)
        lat = co.conv2d_rows(c4, self.latlayer1.weight, self.latlayer1.bias)
        top = co.conv2d_rows(c5, self.toplayer.weight, self.toplayer.bias)
        up = F.interpolate(top.permute(0, 3, 1, 2), size=lat.shape[1:3], mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        return co.conv2d_rows((up + lat).contiguous(), self.smooth1.weight, self.smooth1.bias)


class PoseEstimator2D(_PackedModule):
    """models/pose_estimator_2d.py:10-86"""

    def __init__(self):
        super().__init__()
        self._conv_cache = _PackCache()               # inference launch arguments of `conv` (forge_amd/frozen.py)
        self.backbone = FPN()
        self.cross_attn_layers = 3
        self.self_attn_layers = 3
        self.cross_attn_blks = nn.ModuleList([CrossAttention(4, 256, 256, mlp_ratio=4) for _ in range(3)])
        self.self_attn_blks = nn.ModuleList([SelfAttention(4, 256, mlp_ratio=4) for _ in range(3)])
        lrelu = lambda: nn.LeakyReLU(inplace=True)
        self.conv = nn.Sequential(nn.Conv2d(256, 256, 3, padding=1, stride=2), nn.BatchNorm2d(256), lrelu(),
                                  nn.Conv2d(256, 512, 3, padding=1, stride=2), nn.BatchNorm2d(512), lrelu(),
                                  nn.Conv2d(512, 512, 3, padding=1, stride=2), nn.BatchNorm2d(512), lrelu(),
                                  nn.Conv2d(512, 1024, 3, padding=1, stride=2), nn.BatchNorm2d(1024), lrelu())
        self.out = nn.Sequential(nn.Linear(1024, 256), nn.BatchNorm1d(256), nn.LeakyReLU(), nn.Linear(256, 7))
        # float64 parameter, as in the reference (numpy float64 embedding wrapped in nn.Parameter, :50-51)
        self.pos_emb = nn.Parameter(0.05 * torch.from_numpy(sincos_pos_embed_2d(256, 16))[None])

    def _conv_rows(self, rows):
        """`self.conv` (four Conv2d(k = 3, stride 2) + BatchNorm2d + LeakyReLU, models/pose_estimator_2d.py:36-48) on NHWC rows [n,h,w,256] - what the
        attention blocks produce anyway - through libforge_hip.so (convops.conv2d_rows, bn_act_rows): MIOpen ran these on its naive fp32 kernels."""
        from . import convops as co
        from . import frozen as fz
        from .fusion import bn_act_rows
        mods = list(self.conv)
        if fz.frozen_ok(rows, self.conv) and not any(v % (1 << (len(mods) // 3)) for v in rows.shape[1:3]):
            return fz.run_chain(fz.pack_chain(self._conv_cache, fz.chain_specs(self.conv)), rows[:, None])[:, 0]
        for i in range(0, len(mods), 3):
            conv, bn, act = mods[i:i + 3]
            if rows.shape[1] % 2 or rows.shape[2] % 2:
                raise RuntimeError("forge_amd: PoseEstimator2D.conv needs even feature-map extents at every stride-2 convolution, got %s" % (tuple(rows.shape[1:3]),))
            rows = bn_act_rows(bn, co.conv2d_rows(rows.contiguous(), conv.weight, conv.bias, stride=conv.stride[0]), act.negative_slope)
        return rows

    def forward(self, x, return_features=False):
        """x [B,T,3,H,W] -> pose features [B(T-1),1024] or 7-D pose"""
        B, T, C, H, W = x.shape
        feat = self.backbone.forward_rows(x.reshape(B * T, C, H, W))      # [B*T,h,w,256] NHWC rows (raises on host tensors)
        h2, w2 = feat.shape[1:3]
        feat = feat.reshape(B, T, h2 * w2, 256)                           # [B,T,N,256]
        pos = self.pos_emb.to(feat.device)
        feat_canonical = (feat[:, 0] + pos).to(feat.dtype)                # [B,N,256]
        feat = (feat[:, 1:] + pos.unsqueeze(1)).to(feat.dtype).reshape(B, (T - 1) * h2 * w2, 256)
        for cross, selfa in zip(self.cross_attn_blks, self.self_attn_blks):
            feat = selfa(cross(x_q=feat, x_k=feat_canonical, x_v=feat_canonical, residual=feat))
        feat = self._conv_rows(feat.reshape(B * (T - 1), h2, w2, 256)).reshape(B * (T - 1), -1).squeeze()
        return feat if return_features else self.out(feat)
