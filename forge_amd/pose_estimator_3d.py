"""3-D pose estimator (cross-attention over feature volumes): the reference's attribute surface (`encoder_traj`, `.toSE3`, `.pose_dim`,
`return_features=`) and state_dict keys (`encoder_traj.*`, 26.65 M parameters); architecture re-expressed from
models/pose_estimator_3d.py:9-144 and the Block/Attention/Mlp/positional-embedding helpers of models/model_utils.py:59-256.

On the MI355X its eight 3x3x3 convolutions (stride 1 and 2) + BatchNorm + LeakyReLU run on libforge_hip.so (convops.conv3d_rows, bn_act_rows on
channels-last rows; round 5): in the joint fine-tune step (BASELINE configs[4]) MIOpen served them with its `naive_conv_*` fp32 kernels - 164 ms
of a 256 ms step (profiles/r05_joint_grid32_kernel_share_before.txt). The attention block (1x1 projections, 4096-token softmax, MLP) stays
stock torch (rocBLAS GEMMs). CPU tensors run the same modules on torch's own kernels (the architecture pin of tests/test_oracle_golden.py)."""
import math

import torch
import torch.nn as nn

from . import convops as co
from . import geo_utils

_ROT_DIMS = {"euler": 3, "quat": 4, "6D": 6, "9D": 9}


def sincos_pos_embed_3d(embed_dim, grid_size, temporal_size):
    """models/model_utils.py:59-88: interleaved sin/cos per axis, concatenated (t, w, h), cut to embed_dim.
    Returns [temporal*grid*grid, embed_dim]."""
    ch = int(math.ceil(embed_dim / 6) * 2)
    ch += ch % 2
    inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))

    def axis(n):
        a = torch.arange(n).float()[:, None] * inv_freq[None]
        return torch.stack((a.sin(), a.cos()), dim=-1).flatten(-2)      # [n, ch]

    emb = torch.zeros(temporal_size, grid_size, grid_size, 3 * ch)
    emb[..., :ch] = axis(temporal_size)[:, None, None]
    emb[..., ch:2 * ch] = axis(grid_size)[None, :, None]
    emb[..., 2 * ch:] = axis(grid_size)[None, None, :]
    return emb.reshape(-1, 3 * ch)[:, :embed_dim]


class Attention(nn.Module):
    """models/model_utils.py:207-229 — unscaled dot-product attention, no parameters."""

    def __init__(self, dim, num_heads=1):
        super().__init__()
        self.num_heads = num_heads

    def get_attn(self, query, key):
        return torch.matmul(query, key.transpose(-2, -1)).softmax(dim=-1)

    def forward(self, query, key, value):
        B, N, C = query.shape
        if self.num_heads == 1:
            from . import ops
            if ops.attention_applies(query, key, value):                  # inference on the MI355X: the N x N matrix never leaves registers
                return ops.attention(query, key, value)
        split = lambda x: x.reshape(B, N, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3)
        q, k, v = split(query), split(key), split(value)
        attn = torch.matmul(q, k.transpose(-2, -1)).softmax(dim=-1)
        return torch.matmul(attn, v).transpose(1, 2).reshape(B, N, C)


class Mlp(nn.Module):
    """models/model_utils.py:232-255"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)
        for fc in (self.fc1, self.fc2):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.normal_(fc.bias, std=1e-6)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Block(nn.Module):
    """models/model_utils.py:144-204 — 1x1-conv q/k/v encoders on [B,C,N] tensors, shared LayerNorm for
    q and k, residual attention + MLP."""

    def __init__(self, dim, num_heads=1, mlp_ratio=4.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, return_attn=False):
        super().__init__()
        self.channels = dim
        self.encode_query = nn.Conv1d(dim, dim, 1)
        self.encode_key = nn.Conv1d(dim, dim, 1)
        self.attn = Attention(dim, num_heads=num_heads)
        self.encode_value = nn.Conv1d(dim, dim, 1)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.norm = norm_layer(dim)

    def _qk(self, query, key, query_embed, key_embed):
        q = query if query_embed is None else query + query_embed.to(query)
        k = key if key_embed is None else key + key_embed.to(key)
        q = self.norm(q.permute(0, 2, 1)).permute(0, 2, 1)
        k = self.norm(k.permute(0, 2, 1)).permute(0, 2, 1)
        return self.encode_query(q).permute(0, 2, 1), self.encode_key(k).permute(0, 2, 1)

    # ---- tokens-major twins ([B,N,C] in and out): the 1x1 Conv1d encoders are linear layers over the channels, so with the tokens as rows - what
    # the HIP convolutions around the transformer produce and consume - LayerNorm, encoders, attention and MLP chain without a single permute copy
    @staticmethod
    def _lin(conv, x):
        return torch.nn.functional.linear(x, conv.weight[:, :, 0], conv.bias)

    def qk_tokens(self, query, key):
        return self._lin(self.encode_query, self.norm(query)), self._lin(self.encode_key, self.norm(key))

    def forward_tokens(self, query, key):
        from . import ops
        q, k = self.qk_tokens(query, key)
        v = self._lin(self.encode_value, key)
        x = query + (ops.attention(q, k, v) if ops.attention_applies(q, k, v) else self.attn(query=q, key=k, value=v))
        return x + self.mlp(self.norm2(x))

    def get_attn(self, query, key, query_embed=None, key_embed=None):
        q, k = self._qk(query, key, query_embed, key_embed)
        return self.attn.get_attn(query=q, key=k)                       # [B,N,N]

    def forward(self, query, key, query_embed=None, key_embed=None):
        b = query.shape[0]
        q, k = self._qk(query, key, query_embed, key_embed)
        v = self.encode_value(key).permute(0, 2, 1)
        x = query.permute(0, 2, 1)
        x = x + self.attn(query=q, key=k, value=v)
        x = x + self.mlp(self.norm2(x))
        return x.permute(0, 2, 1).contiguous().view(b, self.channels, -1)


class PoseTransformer(nn.Module):
    """models/pose_estimator_3d.py:116-144"""

    def __init__(self, inp_res=32, dim=64, mlp_ratio=1, coord_dim=64):
        super().__init__()
        self.coord_dim = coord_dim
        self.cross_transformer = Block(dim=dim, mlp_ratio=mlp_ratio, return_attn=True)
        self.self_transformer = Block(dim=dim, mlp_ratio=mlp_ratio, return_attn=False)
        # plain attribute (not a buffer) as in the reference: absent from the state_dict
        self.pos_embed_3d_coord = (sincos_pos_embed_3d(coord_dim, inp_res, inp_res) * 0.1).reshape(1, -1, coord_dim)

    def _pos_embed(self, like):
        """The positional table on `like`'s device / dtype, copied there ONCE per (device, dtype) - the reference copies the host tensor in every
        forward (models/pose_estimator_3d.py:137), a pageable host->device copy that synchronises and cannot be captured into a hipGraph.
        Copied with inference mode OFF, so that a table first built under torch.inference_mode() can still be saved for backward later."""
        cache = self.__dict__.setdefault("_pe_cache", {})
        key = (str(like.device), like.dtype)
        if key not in cache:
            with torch.inference_mode(False):
                cache[key] = self.pos_embed_3d_coord.to(device=like.device, dtype=like.dtype).clone()
        return cache[key]

    def forward_tokens(self, q, k):
        """forward on tokens-major tensors q, k [B,N,C] -> [B,N,C] (inference on the MI355X: PoseEstimator3D._forward_features_hip)."""
        from . import ops
        pe = self._pos_embed(q)
        qn, kn = self.cross_transformer.qk_tokens(q, k)
        coord = ops.attention(qn, kn, pe) if ops.attention_applies(qn, kn, pe) else torch.matmul(self.cross_transformer.attn.get_attn(query=qn, key=kn), pe)
        return self.self_transformer.forward_tokens(coord, coord)

    def forward(self, q, k, q_pe=None, k_pe=None):
        from . import ops
        pe = self._pos_embed(q)
        qn, kn = self.cross_transformer._qk(q, k, None, None)            # [B,N,C] each
        if ops.attention_applies(qn, kn, pe):
            # inference on the MI355X: softmax(q k^T) pe in one launch instead of the [B,N,N] matrix + softmax + matmul
            coord = ops.attention(qn, kn, pe).permute(0, 2, 1)            # [B,C,N]
            return self.self_transformer(query=coord, key=coord)
        attn = self.cross_transformer.attn.get_attn(query=qn, key=kn)    # [B,N,N] (autograd path)
        coord = torch.matmul(attn, pe).permute(0, 2, 1)                  # [B,C,N]
        return self.self_transformer(query=coord, key=coord)


class PoseEstimator3D(co.PackedModule):
    """models/pose_estimator_3d.py:9-113"""

    def __init__(self, config):
        super().__init__()
        self._frozen_cache = co.PackCache()           # inference launch arguments of the four convolution blocks (forge_amd/frozen.py)
        self.rot_representation = config.network.rot_representation
        assert self.rot_representation in _ROT_DIMS
        self.rot_dim = _ROT_DIMS[self.rot_representation]
        self.trans_dim = 3
        self.pose_dim = self.trans_dim + self.rot_dim
        lrelu = lambda: nn.LeakyReLU(inplace=True)
        self.conv3d_1 = nn.Sequential(nn.Conv3d(128, 64, 3, padding=1, stride=2), nn.BatchNorm3d(64), lrelu(),
                                      nn.Conv3d(64, 64, 3, padding=1))
        self.coord_dim = 64
        self.pose_transformer = PoseTransformer(inp_res=16, dim=64, mlp_ratio=2, coord_dim=self.coord_dim)
        self.conv3d_2 = nn.Sequential(nn.Conv3d(64, 64, 3, padding=1), nn.BatchNorm3d(64), lrelu(),
                                      nn.Conv3d(64, 128, 3, padding=1, stride=2), nn.BatchNorm3d(128), lrelu())
        self.conv3d_3 = nn.Sequential(nn.Conv3d(128, 256, 3, padding=1), nn.BatchNorm3d(256), lrelu(),
                                      nn.Conv3d(256, 512, 3, padding=1, stride=2), nn.BatchNorm3d(512), lrelu())
        self.pose_head_1 = nn.Sequential(nn.Conv3d(512, 512, 3, padding=1, stride=2), nn.BatchNorm3d(512), lrelu(),
                                         nn.Conv3d(512, 1024, 3, padding=1, stride=2))
        self.pose_head_2 = nn.Sequential(nn.LayerNorm(1024), lrelu())
        self.out = nn.Sequential(nn.Linear(1024, 256), nn.BatchNorm1d(256), nn.LeakyReLU(),
                                 nn.Linear(256, self.pose_dim + 1))

    @staticmethod
    def _block_rows(seq, rows):
        """An nn.Sequential of Conv3d(k = 3, padding = 1, stride 1 | 2) / BatchNorm3d / LeakyReLU on channels-last rows [n,D,H,W,C]: convolutions
        forward, data and weight gradient on the MFMA implicit-GEMM / wgrad kernels, BatchNorm + the activation behind it as one HIP pass."""
        from . import convops as co
        from .fusion import bn_act_rows
        mods, i = list(seq), 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv3d):
                if m.kernel_size != (3, 3, 3) or m.padding != (1, 1, 1) or m.stride[0] not in (1, 2) or len(set(m.stride)) != 1:
                    raise RuntimeError("forge_amd: PoseEstimator3D expects Conv3d(k=3, padding=1, stride 1|2), got %r" % (m,))
                rows = co.conv3d_rows(rows, m.weight, m.bias, stride=m.stride[0])
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, nn.LeakyReLU):
                    rows, i = bn_act_rows(m, rows, nxt.negative_slope), i + 1
                else:
                    rows = bn_act_rows(m, rows)
            elif isinstance(m, nn.LeakyReLU):
                rows = torch.nn.functional.leaky_relu(rows, m.negative_slope)
            else:
                raise TypeError("unexpected layer in PoseEstimator3D block: %r" % (m,))
            i += 1
        return rows

    def _forward_features_hip(self, features):
        """features [b,t,128,D,H,W] (any strides) on the MI355X -> [b(t-1),1024] through the HIP convolution kernels."""
        b, t, C1, D1, H1, W1 = features.shape
        rows = features.reshape(b * t, C1, D1, H1, W1).permute(0, 2, 3, 4, 1)
        rows = rows if rows.is_contiguous() else rows.contiguous()
        from . import frozen as fz
        if fz.frozen_ok(features, self.conv3d_1, self.conv3d_2, self.conv3d_3, self.pose_head_1):
            # inference (kubric_eval.py predict_initial / demo.py): one launch per convolution, bias + folded BatchNorm + LeakyReLU in its epilogue
            seqs = (self.conv3d_1, self.conv3d_2, self.conv3d_3, self.pose_head_1)
            specs = [fz.chain_specs(q) for q in seqs]
            packed = self._frozen_cache.get([t for sp in specs for t in fz._sources(sp)], lambda: [[fz.pack_layer(*l) for l in sp] for sp in specs])
            block = lambda seq, r: fz.run_chain(packed[seqs.index(seq)], r)
        else:
            block = self._block_rows
        x = block(self.conv3d_1, rows)                                              # [bt,D,H,W,64]
        _, D, H, W, C = x.shape
        if block is not self._block_rows:
            # inference: the rows ARE the tokens - no [B,C,N] round trip around the transformer
            tok = x.reshape(b, t, D * H * W, C)
            ref = tok[:, 0:1].expand(b, t - 1, D * H * W, C).reshape(b * (t - 1), D * H * W, C)
            cur = tok[:, 1:].reshape(b * (t - 1), D * H * W, C)
            rows = self.pose_transformer.forward_tokens(ref, cur).reshape(b * (t - 1), D, H, W, self.coord_dim)
        else:
            x = x.reshape(b, t, D * H * W, C).permute(0, 1, 3, 2)                   # [b,t,C,N] view of the rows
            ref = x[:, 0:1].expand(b, t - 1, C, D * H * W).reshape(b * (t - 1), C, -1)
            cur = x[:, 1:].reshape(b * (t - 1), C, -1)
            x = self.pose_transformer(q=ref, k=cur)                                 # [b(t-1),64,N]
            rows = x.reshape(b * (t - 1), self.coord_dim, D, H, W).permute(0, 2, 3, 4, 1).contiguous()
        rows = block(self.pose_head_1, block(self.conv3d_3, block(self.conv3d_2, rows)))
        if rows.shape[1:4] != (1, 1, 1):                                            # the reference squeezes [n,1024,1,1,1]; other grids have no meaning here
            raise RuntimeError("forge_amd: PoseEstimator3D needs 32^3 feature volumes (pose_head_1 ends at %s)" % (tuple(rows.shape[1:4]),))
        return rows.reshape(b * (t - 1), -1).squeeze()

    def forward(self, features, return_features=False):
        """features [b,t,128,D,H,W] -> (pose [b(t-1),pose_dim], conf [b(t-1),1]) or the 1024-d features"""
        from .fusion import require_hip_input
        require_hip_input("PoseEstimator3D", features)                   # one path; tools/stock_pose.py holds the stock-torch evaluation tests compare against
        x = self.pose_head_2(self._forward_features_hip(features))
        if return_features:
            return x
        x = self.out(x)
        return tuple(x.split([self.pose_dim, 1], dim=-1))

    def toSE3(self, x):
        return {"euler": geo_utils.euler2mat, "quat": geo_utils.quat2mat,
                "6D": geo_utils.rot6d2mat, "9D": geo_utils.rot9d2mat}[self.rot_representation](x)
