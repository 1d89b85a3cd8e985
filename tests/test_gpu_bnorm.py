"""Train-mode BatchNorm (+ activation) on the HIP kernels (csrc/bnorm.hip, fusion.bn_act_rows) against torch.nn.BatchNorm in float64."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,slope", [((2, 4, 6, 8, 128), 0.01), ((3, 16, 20, 8), 0.0), ((1, 3, 5, 7, 2048), 1.0), ((5, 9, 11, 24), 0.01),
                                         ((2, 6, 7, 9, 96), 0.01), ((3, 10, 12, 160), 0.0)])      # C / 4 does not divide 256 (round 5: the block reduction assumed it did)
def test_bn_train_forward_backward_and_running_stats(shape, slope):
    from forge_amd.fusion import bn_act_rows
    dev = torch.device("cuda:0")
    C = shape[-1]
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(*shape, generator=g) * 1.7 + 0.4)
    dy = torch.randn(*shape, generator=g)
    bn = (nn.BatchNorm3d if len(shape) == 5 else nn.BatchNorm2d)(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    ref_bn = copy.deepcopy(bn).double().train()
    nd = len(shape)
    x64 = x.double().requires_grad_(True)
    y64 = ref_bn(x64.permute(0, nd - 1, *range(1, nd - 1))).permute(0, *range(2, nd), 1)
    if slope != 1.0:
        y64 = torch.nn.functional.leaky_relu(y64, slope)
    y64.backward(dy.double())
    hb = bn.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    y = bn_act_rows(hb, xd, slope)
    y.backward(dy.to(dev))
    scale = max(1.0, y64.abs().max().item())
    assert (y.detach().double().cpu() - y64.detach()).abs().max().item() < 2e-6 * scale
    # gradients: an activation argument within rounding of zero may take the other slope: bound the bulk and the outliers separately
    dxe = (xd.grad.double().cpu() - x64.grad).abs()
    assert (dxe > 1e-5 * x64.grad.abs().max()).double().mean().item() < 1e-4 and dxe.norm().item() < 1e-4 * x64.grad.norm().item()
    assert (hb.weight.grad.double().cpu() - ref_bn.weight.grad).abs().max().item() < 2e-5 * max(1.0, ref_bn.weight.grad.abs().max().item())
    assert (hb.bias.grad.double().cpu() - ref_bn.bias.grad).abs().max().item() < 2e-5 * max(1.0, ref_bn.bias.grad.abs().max().item())
    assert (hb.running_mean.double().cpu() - ref_bn.running_mean).abs().max().item() < 1e-6
    assert (hb.running_var.double().cpu() - ref_bn.running_var).abs().max().item() < 1e-5
    assert int(hb.num_batches_tracked) == 1


def test_bn_eval_keeps_the_torch_module_and_single_process_syncbn_runs_hip():
    """Eval mode is not taken by the HIP kernels (same result as the module itself); a SyncBatchNorm module in a single process (no
    process group) normalises with its own batch statistics on the HIP kernels, exactly like nn.BatchNorm3d."""
    from forge_amd.fusion import bn_act_rows
    dev = torch.device("cuda:0")
    x = torch.randn(2, 4, 4, 4, 32, device=dev)
    bn = nn.BatchNorm3d(32).to(dev).eval()
    ref = torch.nn.functional.leaky_relu(bn(x.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1), 0.01)
    assert torch.equal(bn_act_rows(bn, x, 0.01), ref)
    plain, sync = nn.BatchNorm3d(32).to(dev).train(), nn.SyncBatchNorm(32).to(dev).train()
    assert torch.equal(bn_act_rows(sync, x, 0.01), bn_act_rows(plain, x, 0.01))
    assert torch.equal(sync.running_var, plain.running_var)


@pytest.mark.parametrize("M,C,ld", [(32768, 128, 128), (5120, 2048, 2048), (1000, 32, 64), (144, 96, 96), (3001, 160, 192), (7, 8, 8), (262144, 16, 16), (2621440, 3, 3), (100001, 1, 1),
                                    (999, 7, 7), (5, 3, 3), (40000, 3, 5)])      # C % 4 != 0: the flat walk (conv_rgb's 3-channel bias gradient, the density head's 1)
def test_colsum_kernel_vs_float64(M, C, ld):
    """forge_colsum (the bias gradients of the training path; float64 partial sums, fixed order) against a float64 torch sum: 2e-7 of the
    column's absolute sum; a strided view (ld > C) and a deterministic repeat."""
    from forge_amd import convops as co
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + C)
    buf = (torch.randn(M, ld, generator=g) * 3.0 + 0.5).to(dev)
    x = buf[:, :C]                                                        # ld > C: a strided view (copied to dense rows for C % 4 != 0)
    got = co.colsum(x)
    ref = x.double().sum(dim=0)
    scale = x.double().abs().sum(dim=0)
    assert ((got.double() - ref).abs() <= 2e-7 * scale + 1e-30).all()
    assert torch.equal(got, co.colsum(x))
    x3 = buf.reshape(1, M, ld)[..., :C]                                   # [..., C] input shapes
    assert torch.equal(co.colsum(x3), got)


@pytest.mark.parametrize("shape,slope", [((3, 16, 20, 64), 0.0), ((2, 8, 8, 2048), 0.0), ((2, 4, 6, 8, 32), 0.01)])
def test_bn_train_residual_form_forward_backward(shape, slope):
    """y = act(bn(x) + residual) in ONE apply pass each way (the bottleneck tail relu(bn3(conv3) + identity) of the ResNet trunk,
    torchvision Bottleneck.forward): output, d x, d residual, d gamma / d beta and the running statistics against the torch modules in float64."""
    from forge_amd.fusion import bn_act_rows
    dev = torch.device("cuda:0")
    C = shape[-1]
    g = torch.Generator().manual_seed(C + 1)
    x = (torch.randn(*shape, generator=g) * 1.3 + 0.2)
    res = torch.randn(*shape, generator=g)
    dy = torch.randn(*shape, generator=g)
    bn = (nn.BatchNorm3d if len(shape) == 5 else nn.BatchNorm2d)(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    ref_bn = copy.deepcopy(bn).double().train()
    nd = len(shape)
    x64, r64 = x.double().requires_grad_(True), res.double().requires_grad_(True)
    y64 = ref_bn(x64.permute(0, nd - 1, *range(1, nd - 1))).permute(0, *range(2, nd), 1) + r64
    y64 = torch.relu(y64) if slope == 0.0 else torch.nn.functional.leaky_relu(y64, slope)
    y64.backward(dy.double())
    hb = bn.to(dev).train()
    xd, rd = x.to(dev).requires_grad_(True), res.to(dev).requires_grad_(True)
    y = bn_act_rows(hb, xd, slope, residual=rd)
    y.backward(dy.to(dev))
    scale = max(1.0, y64.abs().max().item())
    assert (y.detach().double().cpu() - y64.detach()).abs().max().item() < 2e-6 * scale
    for got, ref in ((xd.grad, x64.grad), (rd.grad, r64.grad)):
        e = (got.double().cpu() - ref).abs()
        assert (e > 1e-5 * ref.abs().max()).double().mean().item() < 1e-4 and e.norm().item() < 1e-4 * ref.norm().item()
    assert (hb.weight.grad.double().cpu() - ref_bn.weight.grad).abs().max().item() < 2e-5 * max(1.0, ref_bn.weight.grad.abs().max().item())
    assert (hb.bias.grad.double().cpu() - ref_bn.bias.grad).abs().max().item() < 2e-5 * max(1.0, ref_bn.bias.grad.abs().max().item())
    assert (hb.running_var.double().cpu() - ref_bn.running_var).abs().max().item() < 1e-5
    assert int(hb.num_batches_tracked) == 1
    # eval mode under autograd with trainable gamma / beta (a fine-tune with frozen statistics): forge_bn_eval_fwd + the sync-backward kernels with zero
    # totals (_BNEvalRows) - output and every gradient against the torch module in float64; the running statistics are read, never updated
    he = copy.deepcopy(hb).eval()
    for p in he.parameters():
        p.grad = None
    rm0, rv0, nbt0 = he.running_mean.clone(), he.running_var.clone(), int(he.num_batches_tracked)
    ref_e = copy.deepcopy(he).cpu().double()
    x64, r64 = x.double().requires_grad_(True), res.double().requires_grad_(True)
    e64 = ref_e(x64.permute(0, nd - 1, *range(1, nd - 1))).permute(0, *range(2, nd), 1) + r64
    e64 = torch.relu(e64) if slope == 0.0 else torch.nn.functional.leaky_relu(e64, slope)
    e64.backward(dy.double())
    xe, rde = x.to(dev).requires_grad_(True), res.to(dev).requires_grad_(True)
    ye = bn_act_rows(he, xe, slope, residual=rde)
    ye.backward(dy.to(dev))
    assert (ye.detach().double().cpu() - e64.detach()).abs().max().item() < 2e-6 * max(1.0, e64.abs().max().item())
    for got, ref in ((xe.grad, x64.grad), (rde.grad, r64.grad)):
        e = (got.double().cpu() - ref).abs()
        assert (e > 1e-5 * ref.abs().max()).double().mean().item() < 1e-4 and e.norm().item() < 1e-4 * ref.norm().item()
    assert (he.weight.grad.double().cpu() - ref_e.weight.grad).abs().max().item() < 2e-5 * max(1.0, ref_e.weight.grad.abs().max().item())
    assert (he.bias.grad.double().cpu() - ref_e.bias.grad).abs().max().item() < 2e-5 * max(1.0, ref_e.bias.grad.abs().max().item())
    assert torch.equal(he.running_mean, rm0) and torch.equal(he.running_var, rv0) and int(he.num_batches_tracked) == nbt0
    # one path: host rows raise instead of running the torch module
    with pytest.raises(RuntimeError, match="no CPU or stock-PyTorch path"):
        bn_act_rows(copy.deepcopy(he).cpu(), x, slope)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride", [(3, 32, 32, 256, 512, 1, 1), (2, 16, 16, 64, 96, 3, 1), (5, 12, 20, 64, 128, 3, 2), (1, 7, 9, 32, 32, 1, 1)])
def test_conv_epilogue_statistics_feed_batchnorm(N, H, W, Cin, Cout, k, stride):
    """The batch statistics of a BatchNorm as a by-product of the producing convolution's GEMM epilogue (forge_conv_igemm `stats`: float64 column
    sums / sums of squares per 32-row block, fixed order): the blocks sum to the output's own float64 sums, and bn_act_rows fed with them equals
    bn_act_rows running its own statistics pass - output, running statistics, and every gradient (the backward is unchanged)."""
    from forge_amd import convops as co
    from forge_amd.fusion import bn_act_rows
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5).to(dev)
    y, st = co.conv2d_rows(x, w, None, stride=stride, want_stats=True)
    if N * H * W >= 2048:                                  # large enough for an un-split launch of the wide kernel: the by-product exists
        assert st.numel() > 0
    if st.numel():                                         # (small problems take a split-K plan: empty stats, the BatchNorm runs its own pass)
        assert st.dtype == torch.float64 and st.shape[1:] == (2, Cout)
        rows = y.reshape(-1, Cout).double()
        assert (st[:, 0].sum(0) - rows.sum(0)).abs().max().item() < 1e-9 * max(1.0, rows.abs().sum(0).max().item())
        assert (st[:, 1].sum(0) - (rows * rows).sum(0)).abs().max().item() < 1e-9 * (rows * rows).sum(0).max().item()
    outs = []
    for use in (True, False):
        bn = nn.BatchNorm2d(Cout).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(Cout, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(Cout, generator=torch.Generator().manual_seed(2)) * 0.2)
        xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yy, ss = co.conv2d_rows(xa, wa, None, stride=stride, want_stats=True)
        z = bn_act_rows(bn, yy, 0.0, stats=ss if use else None)
        (z * torch.linspace(-1, 1, z.numel(), device=dev).reshape(z.shape)).sum().backward()
        outs.append((z.detach(), bn.running_mean.clone(), bn.running_var.clone(), xa.grad, wa.grad, bn.weight.grad, int(bn.num_batches_tracked)))
    a, b = outs
    assert (a[0] - b[0]).abs().max().item() < 2e-6 * max(1.0, b[0].abs().max().item())
    assert (a[1] - b[1]).abs().max().item() < 1e-7 and (a[2] - b[2]).abs().max().item() < 1e-6 and a[6] == b[6] == 1
    for i in (3, 4, 5):
        assert (a[i] - b[i]).abs().max().item() < 2e-5 * max(1e-6, b[i].abs().max().item()), i
