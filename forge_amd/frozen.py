"""Eval-mode launches WITHOUT an autograd graph of the convolution -> BatchNorm -> LeakyReLU chains outside the reconstruction trunk: the two pose
estimators of FORGE's predicted-pose inference (models/model.py:58-96 as driven by kubric_eval.py's predict_initial and demo.py; layer lists in
models/pose_estimator_3d.py:24-60 and models/pose_estimator_2d.py:36-48, 91-136, 237-275).

Each layer is ONE forge_conv_igemm launch (or the three Winograd launches where convops.wino_applies says so) with the convolution's bias, the folded
BatchNorm y = x * scale + shift, the residual add and the activation in its epilogue; packed weights and folded statistics are cached per owning
module (convops.PackCache, dropped on train() / .to() / load_state_dict by convops.PackedModule). The same modules under autograd, or with a
BatchNorm in train mode, take the autograd path (convops.conv3d_rows / conv2d_rows + fusion.bn_act_rows): this file is only the inference schedule."""
import torch
import torch.nn as nn

from . import _lib
from . import convops as co


def frozen_ok(x, *modules):
    """fp32 rows on the MI355X, no autograd graph wanted, every BatchNorm below `modules` in eval mode."""
    if not (x.is_cuda and x.dtype == torch.float32) or torch.is_grad_enabled():
        return False
    return not any(m.training for mod in modules for m in mod.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))


def chain_specs(seq):
    """nn.Sequential of Conv / BatchNorm / LeakyReLU -> [(conv, bn | None, slope)] (slope 1 = no activation behind that convolution)."""
    mods, specs, i = list(seq), [], 0
    while i < len(mods):
        conv = mods[i]
        if not isinstance(conv, (nn.Conv2d, nn.Conv3d)):
            raise TypeError("forge_amd.frozen: expected a convolution at position %d of %r" % (i, seq))
        bn, slope, i = None, 1.0, i + 1
        if i < len(mods) and isinstance(mods[i], nn.modules.batchnorm._BatchNorm):
            bn, i = mods[i], i + 1
        if i < len(mods) and isinstance(mods[i], (nn.LeakyReLU, nn.ReLU)):
            slope, i = (mods[i].negative_slope if isinstance(mods[i], nn.LeakyReLU) else 0.0), i + 1
        specs.append((conv, bn, slope))
    return specs


def _sources(specs):
    src = []
    for conv, bn, _ in specs:
        src += [conv.weight] + ([conv.bias] if conv.bias is not None else [])
        if bn is not None:
            src += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    return src


def pack_layer(conv, bn, slope, force_affine=False):
    """One convolution (+ BatchNorm + activation) as launch arguments. k in {1, 3} with padding k // 2 and one stride for every axis.
    force_affine: give a layer without BatchNorm / activation the affine epilogue anyway (scale 1, shift 0) so that it can take a residual."""
    nd = 3 if isinstance(conv, nn.Conv3d) else 2
    k, s = conv.kernel_size[0], conv.stride[0]
    if (len(set(conv.kernel_size)) != 1 or len(set(conv.stride)) != 1 or k not in (1, 3) or tuple(conv.padding) != (k // 2,) * nd or s not in (1, 2)
            or conv.groups != 1 or tuple(conv.dilation) != (1,) * nd or conv.in_channels % 32 or conv.out_channels % 32):
        raise RuntimeError("forge_amd.frozen: unsupported convolution %r (k in {1,3}, padding k//2, stride 1|2, channels multiples of 32)" % (conv,))
    w = conv.weight.detach()
    if nd == 2:
        wp, taps = co.pack_conv2d_weight(w)
    elif k == 3:
        wp, taps = co.pack_conv3d_weight(w), co.TAPS_3x3x3
    else:
        wp, taps = w.reshape(1, conv.out_channels, conv.in_channels).contiguous(), [(0, 0, 0)]
    bias = None if conv.bias is None else conv.bias.detach().contiguous()
    if bn is not None:
        if bn.running_mean is None:
            raise RuntimeError("forge_amd.frozen: %r has no running statistics to fold" % (bn,))
        sc, sh = co.bn_affine(bn)
    elif slope != 1.0 or force_affine:
        sc = torch.ones(conv.out_channels, dtype=torch.float32, device=w.device)
        sh = torch.zeros(conv.out_channels, dtype=torch.float32, device=w.device)
    else:
        sc = sh = None
    # Winograd weights only where convops.wino_applies can say yes (3-D: Cin >= 64; 2-D: both sides >= WINO2D_MIN_C)
    wino = k == 3 and s == 1 and ((conv.in_channels >= 64) if nd == 3 else min(conv.in_channels, conv.out_channels) >= co.WINO2D_MIN_C)
    U = co.wino_pack_packed(wp) if wino else None
    return {"wp": wp, "taps": taps, "bias": bias, "sc": sc, "sh": sh, "slope": float(slope), "stride": s, "k": k, "nd": nd, "U": U,
            "cin": conv.in_channels, "cout": conv.out_channels}


def pack_chain(cache, specs):
    """[(conv, bn, slope)] -> launch arguments, rebuilt when a parameter / buffer behind them changed."""
    return cache.get(_sources(specs), lambda: [pack_layer(*s) for s in specs])


@_lib.on_tensor_device
def run_layer(L, x, residual=None):
    """x: channels-last rows [n,D,H,W,C] (a 2-D layer: D = 1) -> act((conv(x) + bias) * scale + shift + residual), rows [n,Do,Ho,Wo,Cout]."""
    n, D, H, W, C = x.shape
    if C != L["cin"]:
        raise ValueError("forge_amd.frozen: %d input channels for a convolution of %d" % (C, L["cin"]))
    x = x if x.is_contiguous() else x.contiguous()
    s, Cout = L["stride"], L["cout"]
    f = lambda v: (v - 1) // s + 1                                      # k = 1 / p = 0 and k = 3 / p = 1 alike
    Do, Ho, Wo = (f(D) if L["nd"] == 3 else 1), f(H), f(W)
    out = torch.empty(n, Do, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    if residual is not None and (L["sc"] is None or residual.shape != out.shape):
        raise ValueError("forge_amd.frozen: a residual needs the affine epilogue (pack_layer(force_affine=True)) and the output's shape")
    epi = co.EPI_AFFINE_ACT if L["sc"] is not None else co.EPI_BIAS
    if residual is not None and not residual.is_contiguous():
        residual = residual.contiguous()
    if L["U"] is not None and co.wino_applies(L["taps"], 1, n, D, H, W, C, 0, Cout):
        V = co.wino_input(x, C, C, n, D, H, W)
        Mm = torch.empty(16, n * D * (H // 2) * (W // 2), Cout, dtype=torch.float32, device=x.device)
        co.wino_gemm(V, C, None, 0, L["U"], Mm, n, D, H // 2, W // 2, Cout)
        co.wino_output(Mm, L["bias"], L["sc"], L["sh"], L["slope"], residual, None, None, out, None, None, n, D, H, W, Cout, Cout, epi)
    else:
        co.conv_igemm(x, C, C, None, 0, 0, L["wp"], L["bias"], L["sc"], L["sh"], L["slope"], residual, None, None, out, None,
                      (n, Do, Ho, Wo), (D, H, W), Cout, Cout, L["taps"], istride=s, epilogue=epi)
    return out


def run_chain(layers, x):
    for L in layers:
        x = run_layer(L, x)
    return x


# ------------------------------------------------------------------------------------------------ bottleneck ResNet (the FPN's bottom-up path)
def pack_resnet(cache, conv0, bn0, stages, slope):
    """Stem + bottleneck stages (conv1..3, bn1..3, downsample) -> launch arguments (see Encoder3D._trunk_packed for the encoder's own trunk,
    which additionally relabels layer4's channels for the 2-D -> 3-D lift)."""
    blocks = [blk for stage in stages for blk in stage]
    specs = []
    for blk in blocks:
        specs += [(blk.conv1, blk.bn1, slope), (blk.conv2, blk.bn2, slope), (blk.conv3, blk.bn3, slope)]
        if blk.downsample is not None:
            specs.append((blk.downsample[0], blk.downsample[1], 1.0))

    def build():
        kh, kw = conv0.kernel_size
        Kp = ((kh * kw * conv0.in_channels + 31) // 32) * 32
        w0 = co.pad_cin(conv0.weight.detach().permute(0, 2, 3, 1).reshape(1, conv0.out_channels, -1).contiguous(), Kp)
        packed = []
        for blk in blocks:
            packed.append({"c1": pack_layer(blk.conv1, blk.bn1, slope), "c2": pack_layer(blk.conv2, blk.bn2, slope), "c3": pack_layer(blk.conv3, blk.bn3, slope),
                           "ds": None if blk.downsample is None else pack_layer(blk.downsample[0], blk.downsample[1], 1.0)})
        return {"stem": (w0,) + co.bn_affine(bn0), "blocks": packed, "ends": [len(stage) for stage in stages]}
    return cache.get([conv0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var] + _sources(specs), build)


@_lib.on_tensor_device
def run_resnet(P, conv0, pool, img, slope):
    """img [N,3,H,W] -> the NHWC output rows [N,1,h,w,C] of every stage. Stem: patch gather + one-tap GEMM with BN + activation folded, max-pool
    kernel; bottlenecks: three GEMM launches (+ the downsample's), residual + activation in conv3's epilogue."""
    N, Ci, Hi, Wi = img.shape
    dev = img.device
    kh, kw = conv0.kernel_size
    s0, p0 = conv0.stride[0], conv0.padding[0]
    C0 = conv0.out_channels
    w0, sc0, sh0 = P["stem"]
    Kp = w0.shape[-1]
    Hc, Wc = (Hi + 2 * p0 - kh) // s0 + 1, (Wi + 2 * p0 - kw) // s0 + 1
    patches = torch.empty(N * Hc * Wc, Kp, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().forge_im2col_nchw(_lib.ptr(img.contiguous()), _lib.ptr(patches), N, Ci, Hi, Wi, kh, kw, s0, p0, Kp, _lib.current_stream()),
               "forge_im2col_nchw")
    c0 = torch.empty(N, Hc, Wc, C0, dtype=torch.float32, device=dev)
    co.conv_igemm(patches, Kp, Kp, None, 0, 0, w0, None, sc0, sh0, slope, None, None, None, c0, None, (N, 1, Hc, Wc), (1, Hc, Wc), C0, C0, [(0, 0, 0)],
                  epilogue=co.EPI_AFFINE_ACT)
    pk, ps, pp = pool.kernel_size, pool.stride, pool.padding
    H, W = (Hc + 2 * pp - pk) // ps + 1, (Wc + 2 * pp - pk) // ps + 1
    x = torch.empty(N, 1, H, W, C0, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().forge_maxpool2d_nhwc(_lib.ptr(c0), _lib.ptr(x), N, Hc, Wc, C0, pk, ps, pp, _lib.current_stream()), "forge_maxpool2d_nhwc")
    outs, bi = [], 0
    for n_blocks in P["ends"]:
        for b in P["blocks"][bi:bi + n_blocks]:
            y = run_layer(b["c2"], run_layer(b["c1"], x))
            idn = x if b["ds"] is None else run_layer(b["ds"], x)
            x = run_layer(b["c3"], y, residual=idn)
        bi += n_blocks
        outs.append(x)
    return outs
