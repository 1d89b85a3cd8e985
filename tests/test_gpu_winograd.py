"""Winograd F(2x2, 3x3) x 3-depth-tap path of the fused ConvGRU convolutions (csrc/winograd.hip, forge_wino_*) against a float64
torch convolution, the direct implicit-GEMM kernel and the oracle's fusion."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _conv_wino(x, x2, w, bias, epilogue=0, **kw):
    """x [n,D,H,W,C1] (+ x2 [n,D,H,W,C2]) channels-last rows -> rows [n D H W][Cout] through input transform / point GEMMs / output transform."""
    from forge_amd import convops as co
    n, D, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cout = w.shape[0]
    V1 = co.wino_input(x, C1, C1, n, D, H, W)
    V2 = None if x2 is None else co.wino_input(x2, C2, C2, n, D, H, W)
    R = n * D * (H // 2) * (W // 2)
    Mm = torch.empty(16, R, Cout, device=x.device)
    co.wino_gemm(V1, C1, V2, C2, co.wino_pack_weight(w), Mm, n, D, H // 2, W // 2, Cout)
    out = torch.empty(n * D * H * W, Cout, device=x.device)
    co.wino_output(Mm, bias, kw.get("scale"), kw.get("shift"), kw.get("slope", 1.0), kw.get("residual"), None, None, out, None, None, n, D, H, W, Cout, Cout,
                   epilogue)
    return out


@pytest.mark.parametrize("shape", [(2, 6, 8, 10, 64, 0, 96), (1, 5, 12, 6, 32, 96, 64), (3, 1, 2, 2, 32, 0, 32)])
def test_wino_conv_matches_float64_and_direct_kernel(shape):
    from forge_amd import convops as co
    n, D, H, W, C1, C2, Cout = shape
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, D, H, W, C1 + C2, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) / (27 * (C1 + C2)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 4, 1).reshape(-1, Cout)
    xd = x.to(dev)
    x1 = xd[..., :C1].contiguous()
    x2 = xd[..., C1:].contiguous() if C2 else None
    got = _conv_wino(x1, x2, w.to(dev), bias.to(dev))
    direct = torch.empty_like(got)
    co.conv_igemm(x1, C1, C1, x2, C2, C2, co.pack_conv3d_weight(w.to(dev)), bias.to(dev), None, None, 1.0, None, None, None, direct, None,
                  (n, D, H, W), (D, H, W), Cout, Cout, co.TAPS_3x3x3)
    e_w = (got.double().cpu() - ref).abs().max().item()
    e_d = (direct.double().cpu() - ref).abs().max().item()
    # fp32 rounding only: outputs are O(1); the direct kernel lands at ~1e-6 and Winograd within a small factor of it
    assert e_d < 5e-6 and e_w < 1e-5, (e_w, e_d)
    assert e_w < 4 * e_d + 2e-6, (e_w, e_d)


def test_wino_affine_epilogue_with_residual():
    n, D, H, W, C, Cout = 2, 4, 6, 8, 64, 64
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, D, H, W, C, generator=g)
    w = torch.randn(Cout, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
    bias, scale, shift = (torch.randn(Cout, generator=g) for _ in range(3))
    res = torch.randn(n * D * H * W, Cout, generator=g)
    y = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 4, 1).reshape(-1, Cout)
    ref = torch.nn.functional.leaky_relu((y + res.double()) * scale.double() + shift.double(), 0.01)
    got = _conv_wino(x.to(dev), None, w.to(dev), bias.to(dev), epilogue=1, scale=scale.to(dev), shift=shift.to(dev), slope=0.01, residual=res.to(dev))
    # the HIP epilogue adds the residual to acc + bias before the affine map, as written above
    assert (got.double().cpu() - ref).abs().max().item() < 2e-5


def test_fuse_winograd_matches_direct_kernels_and_oracle(monkeypatch):
    """Encoder3D.fuse on the Winograd path vs the direct implicit-GEMM path (FORGE_WINOGRAD=0) and the CPU oracle."""
    import forge_oracle as fo
    from forge_amd import synthetic as syn
    from forge_amd.fusion import ConvGRU_3D
    dev = _dev()
    gru = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=32, hidden_size=32)
    pre = "encoder_3d.fusion_feature."
    w = syn.seeded_state_dict({pre + k: v for k, v in gru.state_dict().items()}, 9)
    gru.load_state_dict({k[len(pre):]: v for k, v in w.items()})
    gru = gru.to(dev).eval()
    x = torch.randn(2, 3, 32, 6, 8, 10, generator=torch.Generator().manual_seed(4))
    ref = fo.fuse(x, w)
    with torch.no_grad():
        monkeypatch.setenv("FORGE_WINOGRAD", "1")
        a = gru.fuse_hip(x.to(dev)).cpu()
        monkeypatch.setenv("FORGE_WINOGRAD", "0")
        d = gru.fuse_hip(x.to(dev)).cpu()
    assert a.shape == ref.shape
    assert (a - d).abs().max().item() < 2e-5, (a - d).abs().max().item()
    assert (a - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    # odd H: the Winograd path does not apply and fuse_hip keeps the direct kernel
    x2 = torch.randn(1, 2, 32, 4, 5, 6, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        monkeypatch.setenv("FORGE_WINOGRAD", "1")
        o = gru.fuse_hip(x2.to(dev)).cpu()
    assert (o - fo.fuse(x2, w)).abs().max().item() < 1e-4
