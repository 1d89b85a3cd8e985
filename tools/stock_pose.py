"""Stock-torch evaluation of the two pose estimators: TEST / PROBE infrastructure, not a product path.

forge_amd.pose_estimator_{2d,3d} have ONE implementation - the libforge_hip.so convolutions + HIP BatchNorm on the MI355X - and raise on host
tensors. What tests and probes compare that path against (the same module in float64 on the CPU, the same module on torch's own GPU kernels,
the architecture pin of tests/test_oracle_golden.py against the reference's predictions) is the evaluation below: the module's OWN sub-modules
(nn.Conv3d / nn.BatchNorm3d / nn.Sequential containers, the attention blocks) called the way the reference calls them
(models/pose_estimator_3d.py:62-113, models/pose_estimator_2d.py:53-86, 113-136), on any device and dtype.

    from stock_pose import features_2d, features_3d, forward_2d, forward_3d, patched
    with patched(model.encoder_traj, model.encoder_traj_2d):      # probes: the module's forward replaced by the stock evaluation
        ..."""
import contextlib

import torch
import torch.nn.functional as F


def _attention(q, k, v):
    """models/model_utils.py:207-229 with one head: unscaled softmax(q k^T) v."""
    return torch.matmul(torch.matmul(q, k.transpose(-2, -1)).softmax(dim=-1), v)


def _block(blk, query, key):
    """models/model_utils.py:144-204 (Block.forward) on [B,C,N] tensors."""
    b = query.shape[0]
    q, k = blk._qk(query, key, None, None)
    v = blk.encode_value(key).permute(0, 2, 1)
    x = query.permute(0, 2, 1)
    x = x + _attention(q, k, v)
    x = x + blk.mlp(blk.norm2(x))
    return x.permute(0, 2, 1).contiguous().view(b, blk.channels, -1)


def _pose_transformer(pt, q, k):
    """models/pose_estimator_3d.py:135-144."""
    qn, kn = pt.cross_transformer._qk(q, k, None, None)
    attn = torch.matmul(qn, kn.transpose(-2, -1)).softmax(dim=-1)
    coord = torch.matmul(attn, pt.pos_embed_3d_coord.to(q)).permute(0, 2, 1)
    return _block(pt.self_transformer, coord, coord)


def features_3d(mod, features):
    """PoseEstimator3D: features [b,t,128,D,H,W] -> [b(t-1),1024] (models/pose_estimator_3d.py:62-104)."""
    b, t, C1, D1, H1, W1 = features.shape
    x = mod.conv3d_1(features.reshape(b * t, C1, D1, H1, W1))
    _, C, D, H, W = x.shape
    x = x.reshape(b, t, C, D * H * W)
    ref = x[:, 0:1].repeat(1, t - 1, 1, 1).reshape(b * (t - 1), C, -1)
    cur = x[:, 1:].reshape(b * (t - 1), C, -1)
    x = _pose_transformer(mod.pose_transformer, ref, cur).reshape(b * (t - 1), mod.coord_dim, D, H, W)
    x = mod.conv3d_3(mod.conv3d_2(x))
    return mod.pose_head_2(mod.pose_head_1(x).squeeze())


def forward_3d(mod, features, return_features=False):
    x = features_3d(mod, features)
    if return_features:
        return x
    return tuple(mod.out(x).split([mod.pose_dim, 1], dim=-1))


def fpn(bb, x):
    """FPN: [n,3,H,W] -> p4 [n,256,H/16,W/16] (models/pose_estimator_2d.py:113-136)."""
    c4 = bb.layer3(bb.layer2(bb.layer1(bb.layer0(x))))
    c5 = bb.layer4(c4)
    lat = bb.latlayer1(c4)
    p4 = F.interpolate(bb.toplayer(c5), size=lat.shape[-2:], mode="bilinear", align_corners=False) + lat
    return bb.smooth1(p4)


def features_2d(mod, x):
    """PoseEstimator2D: x [B,T,3,H,W] -> [B(T-1),1024] (models/pose_estimator_2d.py:53-86)."""
    B, T, C, H, W = x.shape
    feat = fpn(mod.backbone, x.reshape(B * T, C, H, W))
    h2, w2 = feat.shape[-2:]
    feat = feat.reshape(B, T, 256, h2 * w2).permute(0, 1, 3, 2)
    pos = mod.pos_emb.to(feat.device)
    canon = (feat[:, 0] + pos).to(feat.dtype)
    feat = (feat[:, 1:] + pos.unsqueeze(1)).to(feat.dtype).reshape(B, (T - 1) * h2 * w2, 256)
    for cross, selfa in zip(mod.cross_attn_blks, mod.self_attn_blks):
        feat = selfa(cross(x_q=feat, x_k=canon, x_v=canon, residual=feat))
    feat = feat.reshape(B * (T - 1), h2, w2, 256).permute(0, 3, 1, 2)
    return mod.conv(feat).squeeze()


def forward_2d(mod, x, return_features=False):
    feat = features_2d(mod, x)
    return feat if return_features else mod.out(feat)


def stock_forward(mod):
    """The stock evaluation matching `mod`'s class, as a callable with the module's forward signature."""
    name = type(mod).__name__
    if name == "PoseEstimator3D":
        return lambda features, return_features=False: forward_3d(mod, features, return_features)
    if name == "PoseEstimator2D":
        return lambda x, return_features=False: forward_2d(mod, x, return_features)
    raise TypeError("no stock evaluation for %s" % name)


@contextlib.contextmanager
def patched(*mods):
    """Inside the block `mod(...)` runs the stock evaluation (an instance attribute `forward` shadows the class method)."""
    for m in mods:
        m.forward = stock_forward(m)
    try:
        yield
    finally:
        for m in mods:
            del m.forward
