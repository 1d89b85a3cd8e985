"""Executed-FLOP meter for the matrix-core launches of libforge_hip.so (measurement aid for bench.py / tools; not on the product path).

`with FlopMeter() as m: step()` wraps the ctypes entry points whose work runs on the fp32 MFMA pipe - forge_conv_igemm, forge_wino_gemm,
forge_conv_wgrad, forge_wino_wgrad, forge_attention_fwd - for the duration of the block and sums the FLOPs each launch EXECUTES, computed from the call's own
arguments (2 M N taps Cin for a direct / data-gradient / weight-gradient convolution, 2 x 16 R N kd Cin for the 16 Winograd point problems).
Only eager launches made by this process are seen (a hipGraph replay makes no Python calls): meter one eager pass, time the replay.
"""
import torch

from . import _lib


def _v(a):
    return a.value if hasattr(a, "value") else a


def _igemm(a):          # forge_conv_igemm(in1,C1,ld1,bs1,in2,C2,ld2,bs2,wp,bias,scale,shift,slope,residual,aux_h,aux_z,out,out2,out3,n,D,H,W,is,Di,Hi,Wi,Cout,ldo,taps,ntaps,...)
    C1, C2 = _v(a[1]), _v(a[5])
    n, D, H, W, Cout, ntaps = _v(a[19]), _v(a[20]), _v(a[21]), _v(a[22]), _v(a[27]), _v(a[30])
    return 2.0 * n * D * H * W * Cout * ntaps * (C1 + C2)          # merged transposed-conv phases: M rows x ntaps / P taps x P phases = the same product


def _wino_gemm(a):      # forge_wino_gemm(V1,C1,ld1,bs1,pt1,V2,C2,ld2,bs2,pt2,U,Mm,n,D,Ht,Wt,Cout,kd,tile,stream) / forge_wino_gemm_half (same, no tile)
    return 2.0 * 16 * _v(a[12]) * _v(a[13]) * _v(a[14]) * _v(a[15]) * _v(a[16]) * _v(a[17]) * (_v(a[1]) + _v(a[6]))


def _wgrad(a):          # forge_conv_wgrad(dy,ldy,x1,C1,ld1,bs1,x2,C2,ld2,bs2,dwp,n,D,H,W,is,Di,Hi,Wi,Cout,taps,ntaps,stream)
    return 2.0 * _v(a[11]) * _v(a[12]) * _v(a[13]) * _v(a[14]) * _v(a[19]) * _v(a[21]) * (_v(a[3]) + _v(a[7]))


def _wino_wgrad(a):     # forge_wino_wgrad(dMm,V1,C1,bs1,pt1,V2,C2,bs2,pt2,dU,n,D,Ht,Wt,Cout,kd,stream)
    return 2.0 * 16 * _v(a[10]) * _v(a[11]) * _v(a[12]) * _v(a[13]) * _v(a[14]) * _v(a[15]) * (_v(a[2]) + _v(a[6]))


def _attention(a):      # forge_attention_fwd(q,k,v,v_batch_rows,out,B,Nq,Nk,d,stream): q k^T and p v
    return 4.0 * _v(a[5]) * _v(a[6]) * _v(a[7]) * _v(a[8])


_ENTRIES = {"forge_conv_igemm": _igemm, "forge_wino_gemm": _wino_gemm, "forge_wino_gemm_half": _wino_gemm, "forge_conv_wgrad": _wgrad, "forge_wino_wgrad": _wino_wgrad,
            "forge_attention_fwd": _attention}


class FlopMeter:
    def __init__(self):
        self.flops = {k: 0.0 for k in _ENTRIES}
        self.launches = {k: 0 for k in _ENTRIES}

    def __enter__(self):
        self._lib = _lib.lib()
        self._orig = {}
        for name, fn in _ENTRIES.items():
            orig = getattr(self._lib, name)
            self._orig[name] = orig

            def wrapped(*a, _orig=orig, _fn=fn, _name=name):
                self.flops[_name] += _fn(a)
                self.launches[_name] += 1
                return _orig(*a)
            setattr(self._lib, name, wrapped)          # instance attribute of the CDLL handle: what `_lib.lib().<name>` resolves to
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(self._lib, name, orig)
        return False

    @property
    def gflop(self):
        return sum(self.flops.values()) / 1e9

    def summary(self):
        return {"executed_gflop": self.gflop, "launches": dict(self.launches), "gflop_by_entry": {k: v / 1e9 for k, v in self.flops.items()}}


def stage_replay_ms(model, sample, dev, iters=20):
    from . import geo_utils
    from .graph import GraphedCall
    """{stage: ms per replay} for FORGE's inference step on `sample` (b scenes, 5 input views, V cameras)."""
    e3 = model.encoder_3d
    b = sample["images"].shape[0]
    t = 5
    with torch.no_grad():
        img = sample["images"][:, :t].reshape(b * t, 3, 256, 256).contiguous()
        lifted = e3._trunk_hip(img)
        feats = e3._conv1_hip(lifted).reshape(b, t, 128, 32, 32, 32)
        poses = sample["cam_poses_cv2_canonicalized"][:, :t].contiguous()
        rotated = model.rotate(voxels=feats, camPoses_cv2=poses, grid_size=32, order="distance")
        fused = e3.fuse(rotated)
        fv, dv = e3.heads(fused)
        V = sample["K_cv2"].shape[1]
        cams = geo_utils.camera_dict(sample["cam_extrinsics_cv2_canonicalized"], sample["K_cv2"])
        v2v = model._view2vol(b, V, dev)
    stages = {
        "encoder_resnet": lambda: e3._trunk_hip(img),
        "encoder_conv1": lambda: e3._conv1_hip(lifted),
        "rotate": lambda: model.rotate(voxels=feats, camPoses_cv2=poses, grid_size=32, order="distance"),
        "fuse": lambda: e3.fuse(rotated),
        "heads": lambda: e3.heads(fused),
        "render(+conv_rgb)": lambda: model.render(cams, fv, dv, return_origin_proj=True, view2vol=v2v),
    }
    out = {}
    for name, fn in stages.items():
        def call(fn=fn):
            with torch.no_grad():
                return fn()
        g = GraphedCall(call, dev, warmup=2)
        g()
        torch.cuda.synchronize()
        a, bq = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            g()
        bq.record()
        torch.cuda.synchronize()
        out[name] = a.elapsed_time(bq) / iters
        del g
    return out
