"""Winograd F(2x2, 3x3) x 3-depth-tap path of the fused ConvGRU convolutions (csrc/winograd.hip, forge_wino_*) against a float64
torch convolution, the direct implicit-GEMM kernel and the oracle's fusion."""
import os
import sys

import pytest
import torch

from forge_amd import convops as _co

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
    """Encoder3D.fuse on the Winograd path vs the direct implicit-GEMM path (convops.winograd(False)) and the CPU oracle."""
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
        monkeypatch.setattr(_co.STATE, "winograd", True)
        a = gru.fuse_hip(x.to(dev)).cpu()
        monkeypatch.setattr(_co.STATE, "winograd", False)
        d = gru.fuse_hip(x.to(dev)).cpu()
    assert a.shape == ref.shape
    assert (a - d).abs().max().item() < 2e-5, (a - d).abs().max().item()
    assert (a - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    # odd H: the Winograd path does not apply and fuse_hip keeps the direct kernel
    x2 = torch.randn(1, 2, 32, 4, 5, 6, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        monkeypatch.setattr(_co.STATE, "winograd", True)
        o = gru.fuse_hip(x2.to(dev)).cpu()
    assert (o - fo.fuse(x2, w)).abs().max().item() < 1e-4


def test_wino_weight_kernel_matches_float64_einsum():
    """forge_wino_weights (forward and data-gradient forms) against G w G^T evaluated with a float64 einsum."""
    from forge_amd import convops as co
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    w = torch.randn(40, 24, 3, 3, 3, generator=g)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]], dtype=torch.float64)
    ref = torch.einsum("ia,jb,ockab->ijkoc", G, G, w.double()).reshape(16, 3, 40, 24).float()
    wp = co.pack_conv3d_weight(w.to(dev))
    assert torch.equal(co.wino_pack_packed(wp).cpu(), ref)
    wt = w.flip(2, 3, 4).transpose(0, 1)                      # data gradient = correlation with the flipped kernel, channel roles swapped
    ref_t = torch.einsum("ia,jb,ockab->ijkoc", G, G, wt.double()).reshape(16, 3, 24, 40).float()
    assert torch.equal(co.wino_pack_packed(wp, transpose=True).cpu(), ref_t)


def test_conv3_launch_forward_and_data_gradient_vs_torch():
    """convops.conv3_launch (the dispatcher of the training / refinement paths): forward with a residual and the data gradient, Winograd
    and direct kernels, against torch autograd in float64."""
    from forge_amd import convops as co
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    n, D, H, W, Ci, Co = 2, 4, 6, 8, 128, 64
    x = torch.randn(n, D, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (27 * Ci) ** 0.5
    res = torch.randn(n, D, H, W, Co, generator=g)
    dy = torch.randn(n, D, H, W, Co, generator=g)
    x64 = x.double().permute(0, 4, 1, 2, 3).requires_grad_(True)
    y64 = torch.nn.functional.conv3d(x64, w.double(), padding=1)
    y64.backward(dy.double().permute(0, 4, 1, 2, 3))
    ref_y = y64.detach().permute(0, 2, 3, 4, 1) + res.double()
    ref_dx = x64.grad.permute(0, 2, 3, 4, 1)
    wp = co.pack_conv3d_weight(w.to(dev))
    assert co.wino_applies(co.TAPS_3x3x3, 1, n, D, H, W, Ci, 0, Co)
    for mode in ("1", "0"):
        with co.winograd(mode == "1"):
            y = torch.empty(n, D, H, W, Co, device=dev)
            co.conv3_launch(x.to(dev), Ci, None, 0, wp, None, y, (n, D, H, W), Co, residual=res.to(dev))
            dx = torch.empty(n, D, H, W, Ci, device=dev)
            # Co = 64 < 128 input channels of the data-gradient problem: that one stays on the direct kernel in both modes; use a
            # 128-channel dy as well so that the Winograd data gradient is exercised
            co.conv3_launch(dy.to(dev), Co, None, 0, wp, None, dx, (n, D, H, W), Ci, dgrad=True)
        assert (y.double().cpu() - ref_y).abs().max().item() < 1e-5, mode
        assert (dx.double().cpu() - ref_dx).abs().max().item() < 1e-5, mode
    # wide data gradient (Co = 128 -> Winograd applies)
    w2 = torch.randn(128, 128, 3, 3, 3, generator=g) / (27 * 128) ** 0.5
    dy2 = torch.randn(n, D, H, W, 128, generator=g)
    x2 = torch.zeros(n, 128, D, H, W, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x2, w2.double(), padding=1).backward(dy2.double().permute(0, 4, 1, 2, 3))
    dx2 = torch.empty(n, D, H, W, 128, device=dev)
    co.conv3_launch(dy2.to(dev), 128, None, 0, co.pack_conv3d_weight(w2.to(dev)), None, dx2, (n, D, H, W), 128, dgrad=True)
    assert (dx2.double().cpu() - x2.grad.permute(0, 2, 3, 4, 1)).abs().max().item() < 1e-5


def test_frozen_fusion_winograd_matches_direct(monkeypatch):
    """_FuseFrozen (pose refinement: fused forward, hand-written data-gradient backward) on the Winograd launches vs the direct kernels:
    output and input gradient (relative L2; LeakyReLU / gate nonlinearities amplify nothing here: both run the same tails)."""
    from forge_amd import synthetic as syn
    from forge_amd.fusion import ConvGRU_3D
    dev = _dev()
    gru = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=128, hidden_size=128)
    gru.load_state_dict(syn.seeded_state_dict(gru.state_dict(), 3))
    gru = gru.to(dev).eval()
    for p_ in gru.parameters():
        p_.requires_grad_(False)
    x = (torch.randn(1, 3, 128, 8, 8, 8, generator=torch.Generator().manual_seed(2)) * 0.5).to(dev)
    wgt = torch.randn(1, 128, 8, 8, 8, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setattr(_co.STATE, "winograd", mode == "1")
        xi = x.clone().requires_grad_(True)
        out = gru.fuse_frozen_hip(xi)
        (out * wgt).sum().backward()
        res[mode] = (out.detach(), xi.grad)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(res["1"][0], res["0"][0]) < 1e-5
    assert rel(res["1"][1], res["0"][1]) < 1e-3        # sign flips of LeakyReLU arguments within 1e-6 of zero move single elements (test_gpu_configs)


def test_frozen_fusion_hoisted_reference_view_equals_the_plain_frozen_fusion():
    """_FuseFrozen with const0 (pose refinement: view 0 is the fixed, un-warped reference view of frozen features - its input-half point
    products are made once and added inside step 0's inverse transforms, whose GEMMs contract the hidden-state half only): over three
    "iterations" that change views 1.. but not view 0, and two scenes, output and input gradient equal the plain skip_dx0 form to fp32
    rounding (only the order of additions differs); the dict is filled by the first call and reused."""
    from forge_amd import synthetic as syn
    from forge_amd.fusion import ConvGRU_3D
    dev = _dev()
    gru = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=128, hidden_size=128)
    gru.load_state_dict(syn.seeded_state_dict(gru.state_dict(), 3))
    gru = gru.to(dev).eval()
    for p_ in gru.parameters():
        p_.requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(2, 1, 128, 8, 8, 8, generator=g) * 0.5).to(dev)
    wgt = torch.randn(2, 128, 8, 8, 8, generator=g).to(dev)
    const0 = {}
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for it in range(3):
        rest = (torch.randn(2, 3, 128, 8, 8, 8, generator=g) * 0.5).to(dev)
        outs = []
        for c in (const0, None):
            xi = torch.cat([x0, rest], dim=1).requires_grad_(True)
            out = gru.fuse_frozen_hip(xi, skip_dx0=True, const0=c)
            (out * wgt).sum().backward()
            outs.append((out.detach(), xi.grad))
        if it == 0:
            assert set(const0) == {"MXg0", "MXc0"}
            kept = (const0["MXg0"].data_ptr(), const0["MXg0"].clone())
        assert const0["MXg0"].data_ptr() == kept[0] and torch.equal(const0["MXg0"], kept[1])      # computed once, never rewritten
        assert rel(outs[0][0], outs[1][0]) < 2e-6, it
        assert rel(outs[0][1], outs[1][1]) < 1e-3, it      # (LeakyReLU arguments within rounding of zero may take the other slope, as above)


@pytest.mark.parametrize("n,D,H,W,C", [(2, 3, 8, 12, 128), (1, 1, 32, 32, 512), (3, 5, 6, 4, 36)])
def test_wino_input_dy_equals_the_two_separate_transforms(n, D, H, W, C):
    """forge_wino_input_dy (one pass over an upstream gradient for both of its Winograd forms) == forge_wino_input and forge_wino_dy, bit for bit."""
    from forge_amd import _lib
    from forge_amd import convops as co
    dev = _dev()
    dy = torch.randn(n * D * H * W, C, generator=torch.Generator().manual_seed(C)).to(dev)
    V, dM = co.wino_input_dy(dy, C, n, D, H, W)
    Vref = co.wino_input(dy, C, C, n, D, H, W)
    dMref = torch.empty_like(dM)
    _lib.check(_lib.lib().forge_wino_dy(_lib.ptr(dy), C, _lib.ptr(dMref), n, D, H, W, C, _lib.current_stream()), "forge_wino_dy")
    assert torch.equal(V, Vref) and torch.equal(dM, dMref)


def test_wino_weight_gradient_vs_float64_and_direct_kernel(monkeypatch):
    """convops.conv3_wgrad (dMm = A dy A^T, 16 batched wgrad problems, G^T dU G) against float64 autograd and the direct wgrad kernel:
    two-input form with a batch-strided first operand (the GRU cells' layout) and a single-input form."""
    from forge_amd import convops as co
    dev = _dev()
    g = torch.Generator().manual_seed(17)
    n, D, H, W, C1, C2, Co = 2, 4, 6, 8, 128, 128, 64
    xs = torch.randn(n, 2, D, H, W, C1, generator=g)                       # views stacked: x1 = xs[:, 1] has a batch stride
    x2 = torch.randn(n, D, H, W, C2, generator=g)
    dy = torch.randn(n, D, H, W, Co, generator=g)
    w = torch.zeros(Co, C1 + C2, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    xin = torch.cat([xs[:, 1], x2], dim=-1).double().permute(0, 4, 1, 2, 3)
    torch.nn.functional.conv3d(xin, w, padding=1).backward(dy.double().permute(0, 4, 1, 2, 3))
    ref = w.grad.reshape(Co, C1 + C2, 27).permute(2, 0, 1)                  # packed layout [27][Co][Ci]
    xsd, x2d, dyd = xs.to(dev), x2.to(dev), dy.to(dev)
    x1d = xsd[:, 1]
    monkeypatch.setattr(_co.STATE, "winograd", True)
    assert co.wino_wgrad_applies(n, D, H, W, C1, C2, Co)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setattr(_co.STATE, "winograd", mode == "1")
        dwp = torch.zeros(27, Co, C1 + C2, device=dev)
        co.conv3_wgrad(dyd, x1d, C1, x2d, C2, dwp, (n, D, H, W), Co, bs1=co._batch_stride_rows(x1d))
        out[mode] = dwp.double().cpu()
    scale = ref.abs().max().item()
    e_w, e_d = (out["1"] - ref).abs().max().item() / scale, (out["0"] - ref).abs().max().item() / scale
    assert e_d < 2e-5 and e_w < 2e-5, (e_w, e_d)
    # single input, Cout = 256 (the data-gradient-shaped problem of the gates convolution)
    x = torch.randn(1, 4, 8, 8, 256, generator=g)
    dy2 = torch.randn(1, 4, 8, 8, 256, generator=g)
    w2 = torch.zeros(256, 256, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x.double().permute(0, 4, 1, 2, 3), w2, padding=1).backward(dy2.double().permute(0, 4, 1, 2, 3))
    ref2 = w2.grad.reshape(256, 256, 27).permute(2, 0, 1)
    monkeypatch.setattr(_co.STATE, "winograd", True)
    dwp = torch.zeros(27, 256, 256, device=dev)
    co.conv3_wgrad(dy2.to(dev), x.to(dev), 256, None, 0, dwp, (1, 4, 8, 8), 256)
    assert (dwp.double().cpu() - ref2).abs().max().item() < 2e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("hw", [(200, 200), (72, 104), (64, 96)])
def test_encoder_odd_and_even_trunk_grids_vs_oracle(hw):
    """Encoder3D.get_feat3D at image sizes whose /8 trunk grid is odd (25 x 25, 9 x 13: the 2-D Winograd launches of layer3/4 and of conv1
    decline and the direct kernel runs) and even (8 x 12: Winograd) against the oracle - 5e-6 of the output's max either way."""
    import forge_oracle as fo
    from forge_amd import synthetic as syn
    from forge_amd.encoder import Encoder3D
    dev = _dev()
    enc = Encoder3D(syn.kubric_config())
    w = syn.seeded_state_dict({"encoder_3d." + k: v for k, v in enc.state_dict().items()}, 0)
    enc.load_state_dict({k[len("encoder_3d."):]: v for k, v in w.items()})
    enc = enc.to(dev).eval()
    img = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        got = enc.get_feat3D(img.to(dev)).cpu()
        ref = fo.get_feat3D(img, w)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 5e-6 * ref.abs().max().item()


def test_input_transform_of_the_view_mean_equals_mean_then_transform():
    """forge_wino_input with nsum views (the mean of models/encoder.py:62 taken inside the transform) == torch.mean over the views followed by
    the plain transform, bit for bit (sum in view order, then / nsum)."""
    from forge_amd import convops as co
    dev = _dev()
    b, t, D, H, W, C = 2, 5, 3, 8, 12, 32
    x = torch.randn(b, t, D, H, W, C, generator=torch.Generator().manual_seed(8)).to(dev)
    vol = D * H * W
    got = co.wino_input(x, C, C, b, D, H, W, bs=t * vol, nsum=t, sum_stride=vol)
    acc = x[:, 0].clone()
    for k in range(1, t):
        acc = acc + x[:, k]
    ref = co.wino_input((acc / float(t)).contiguous(), C, C, b, D, H, W)
    assert torch.equal(got, ref)
    ref_mean = co.wino_input(x.mean(dim=1).contiguous(), C, C, b, D, H, W)          # torch's own reduction order: equal up to rounding
    assert (got - ref_mean).abs().max().item() < 1e-5
    sub = co.wino_input(x[:, 1:], C, C, b, D, H, W, bs=t * vol, nsum=3, sum_stride=vol)    # a run of views 1..3
    ref3 = co.wino_input(((x[:, 1] + x[:, 2] + x[:, 3]) / 3.0).contiguous(), C, C, b, D, H, W)
    assert torch.equal(sub, ref3)


@pytest.mark.parametrize("shape", [(1, 8, 32, 32, 64, 64, 128), (1, 9, 32, 30, 32, 32, 96), (2, 32, 32, 32, 128, 0, 256), (1, 8, 32, 32, 64, 0, 128)])
def test_half_inverse_transform_in_the_gemm_epilogue_is_bitwise_the_two_launch_form(shape):
    """forge_wino_gemm_half + forge_wino_output_half (the inverse transform's row stage in the GEMM epilogue: four points per workgroup, 8 planes through
    HBM instead of 16) against forge_wino_gemm + forge_wino_output: the same fp32 operations in the same order, so every epilogue must agree BIT FOR BIT -
    full and ragged tile rows / output columns, one and two concatenated operands, several batch elements (depth taps must not cross them)."""
    from forge_amd import convops as co
    n, D, H, W, C1, C2, Cout = shape
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(5)
    x1 = torch.randn(n, D, H, W, C1, device=dev, generator=g)
    x2 = torch.randn(n, D, H, W, C2, device=dev, generator=g) if C2 else None
    U = co.wino_pack_weight(torch.randn(Cout, C1 + C2, 3, 3, 3, device=dev, generator=g) * 0.05)
    R, M = n * D * (H // 2) * (W // 2), n * D * H * W
    assert co.wino_half_applies(R, Cout, C1 + C2)
    V1 = co.wino_input(x1, C1, C1, n, D, H, W)
    V2 = None if x2 is None else co.wino_input(x2, C2, C2, n, D, H, W)
    Mm, Mm8 = torch.empty(16, R, Cout, device=dev), torch.full((8, R, Cout), float("nan"), device=dev)
    co.wino_gemm(V1, C1, V2, C2, U, Mm, n, D, H // 2, W // 2, Cout, half=False)
    co.wino_gemm(V1, C1, V2, C2, U, Mm8, n, D, H // 2, W // 2, Cout, half=True)
    s0 = (Mm[0:4] + Mm[4:8]) + Mm[8:12]                              # rows of A^T M over the point index i (p = 4 i + j)
    s1 = (Mm[4:8] - Mm[8:12]) - Mm[12:16]
    assert torch.equal(Mm8[0:4], s0) and torch.equal(Mm8[4:8], s1)
    bias, sc, sh = (torch.randn(Cout, device=dev, generator=g) for _ in range(3))
    res = torch.randn(M, Cout, device=dev, generator=g)
    for epi, kw in ((co.EPI_BIAS, {}), (co.EPI_AFFINE_ACT, dict(scale=sc, shift=sh, slope=0.01, residual=res))):
        a, b = torch.empty(M, Cout, device=dev), torch.empty(M, Cout, device=dev)
        co.wino_output(Mm, bias, kw.get("scale"), kw.get("shift"), kw.get("slope", 1.0), kw.get("residual"), None, None, a, None, None, n, D, H, W, Cout, Cout, epi,
                       half=False)
        co.wino_output(Mm8, bias, kw.get("scale"), kw.get("shift"), kw.get("slope", 1.0), kw.get("residual"), None, None, b, None, None, n, D, H, W, Cout, Cout, epi,
                       half=True)
        assert torch.equal(a, b)
    # a second addend (the shared input halves of the grouped fusions) in the same 8-plane form: row-combined before the addition - not the same
    # order of fp32 additions as the 16-plane kernel's (m + m2 first), so equal to rounding only
    Mx = torch.randn(16, R, Cout, device=dev, generator=g)
    Mx8 = torch.cat([(Mx[0:4] + Mx[4:8]) + Mx[8:12], (Mx[4:8] - Mx[8:12]) - Mx[12:16]])
    a, b = torch.empty(M, Cout, device=dev), torch.empty(M, Cout, device=dev)
    co.wino_output(Mm, bias, None, None, 1.0, None, None, None, a, None, None, n, D, H, W, Cout, Cout, co.EPI_BIAS, Mm2=Mx, half=False)
    co.wino_output(Mm8, bias, None, None, 1.0, None, None, None, b, None, None, n, D, H, W, Cout, Cout, co.EPI_BIAS, Mm2=Mx8, half=True)
    assert (a - b).abs().max().item() <= 4e-6 * a.abs().max().item()
    if Cout % 2 == 0:                                                 # the GRU tails: gates (z | r -> h r) on Cout = 2 Ch columns, state update on Cout columns
        Ch = Cout // 2
        h = torch.randn(M, Ch, device=dev, generator=g)
        outs = []
        for half in (False, True):
            z, hr, r = (torch.empty(M, Ch, device=dev) for _ in range(3))
            co.wino_output(Mm8 if half else Mm, bias, None, None, 1.0, None, h, None, z, hr, r, n, D, H, W, Cout, Ch, co.EPI_GRU_GATES, half=half)
            outs.append((z, hr, r))
        assert all(torch.equal(u, v) for u, v in zip(*outs))
    hfull, zfull = torch.randn(M, Cout, device=dev, generator=g), torch.rand(M, Cout, device=dev, generator=g)
    outs = []
    for half in (False, True):
        hn, hb, cand = (torch.empty(M, Cout, device=dev) for _ in range(3))
        co.wino_output(Mm8 if half else Mm, bias, sc, sh, 1.0, None, hfull, zfull, hn, hb, cand, n, D, H, W, Cout, Cout, co.EPI_GRU_OUT, half=half)
        outs.append((hn, hb, cand))
    assert all(torch.equal(u, v) for u, v in zip(*outs))


def test_half_form_one_depth_tap_2d_launches_bitwise():
    """The 2-D form (kd = 1: the ResNet layer3 / layer4 3x3 convolutions at many images, planes of the (n, D) grid do not mix) through the four-point
    GEMM: bitwise the 16-plane pair, ragged tile rows included."""
    from forge_amd import convops as co
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(9)
    n, H, W, C, Cout = 37, 16, 16, 256, 256                            # R = 37 x 64 = 2368 tile rows: not a multiple of the 64-row tile
    R, M = n * (H // 2) * (W // 2), n * H * W
    assert co.wino_half_applies(R, Cout, C)
    V = torch.randn(16, R, C, device=dev, generator=g)
    U = torch.randn(16, 1, Cout, C, device=dev, generator=g) * 0.05
    sc, sh = torch.rand(Cout, device=dev, generator=g) + 0.5, torch.randn(Cout, device=dev, generator=g)
    Mm, Mm8 = torch.empty(16, R, Cout, device=dev), torch.empty(8, R, Cout, device=dev)
    co.wino_gemm(V, C, None, 0, U, Mm, n, 1, H // 2, W // 2, Cout, half=False)
    co.wino_gemm(V, C, None, 0, U, Mm8, n, 1, H // 2, W // 2, Cout, half=True)
    ref = torch.einsum("prc,poc->pro", V.double(), U[:, 0].double())
    assert (Mm.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    a, b = torch.empty(M, Cout, device=dev), torch.empty(M, Cout, device=dev)
    co.wino_output(Mm, None, sc, sh, 0.0, None, None, None, a, None, None, n, 1, H, W, Cout, Cout, co.EPI_AFFINE_ACT, half=False)
    co.wino_output(Mm8, None, sc, sh, 0.0, None, None, None, b, None, None, n, 1, H, W, Cout, Cout, co.EPI_AFFINE_ACT, half=True)
    assert torch.equal(a, b)
