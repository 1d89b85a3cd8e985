"""The algebra behind csrc/winograd.hip, checked on the CPU in float64 with plain torch einsums (no HIP): forward, data-gradient and
weight-gradient forms of Winograd F(2x2, 3x3) x 3 depth taps against torch's convolution and autograd. The kernels hard-code exactly these
matrices and index conventions (tile row r = ((n D + z) H/2 + th) W/2 + tw, point p = 4 i + j, patch rows 2 th - 1 .., cols 2 tw - 1 ..)."""
import torch

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]], dtype=torch.float64)


def _input_transform(x):
    """x [n,C,D,H,W] -> V [4,4,n,C,D+2,H/2,W/2] (depth zero-padded by one plane each side)."""
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1))
    d = xp.unfold(3, 4, 2).unfold(4, 4, 2)                     # [n,C,D+2,Ht,Wt,4,4]: rows 2th-1.., cols 2tw-1.. of the unpadded volume
    return torch.einsum("ia,jb,nczuvab->ijnczuv", BT, BT, d)


def _wino_forward(x, w):
    n, C, D, H, W = x.shape
    V = _input_transform(x)
    U = torch.einsum("ia,jb,ockab->ijkoc", G, G, w)            # [4,4,3,Cout,Cin]
    Mm = sum(torch.einsum("ijoc,ijnczuv->ijnozuv", U[:, :, k], V[:, :, :, :, k:k + D]) for k in range(3))
    Y = torch.einsum("pi,qj,ijnozuv->nozupvq", AT, AT, Mm)     # [n,Cout,D,Ht,2,Wt,2]
    return Y.reshape(n, w.shape[0], D, H, W), V


def _setup(seed=0, n=2, C=5, Co=4, D=3, H=4, W=6):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, C, D, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C, 3, 3, 3, generator=g, dtype=torch.float64)
    dy = torch.randn(n, Co, D, H, W, generator=g, dtype=torch.float64)
    return x, w, dy


def test_forward_identity():
    x, w, _ = _setup()
    y, _ = _wino_forward(x, w)
    assert (y - torch.nn.functional.conv3d(x, w, padding=1)).abs().max().item() < 1e-12


def test_data_gradient_is_the_forward_form_with_flipped_transposed_weights():
    """forge_wino_weights(transpose=1): U from w'[ci][co][kd][a][b] = w[co][ci][2-kd][2-a][2-b]; dx = winograd_conv(dy, w')."""
    x, w, dy = _setup(1)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.conv3d(xr, w, padding=1).backward(dy)
    wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()
    dx, _ = _wino_forward(dy, wt)
    assert (dx - xr.grad).abs().max().item() < 1e-12


def test_weight_gradient_in_the_winograd_domain():
    """dMm = A dy A^T per tile (forge_wino_dy), dU[p][kd] = sum_r dMm[p][r] (x) V[p][r + kd plane] (forge_wino_wgrad), dw = G^T dU G (forge_wino_dw)."""
    x, w, dy = _setup(2)
    wr = w.clone().requires_grad_(True)
    torch.nn.functional.conv3d(x, wr, padding=1).backward(dy)
    n, C, D, H, W = x.shape
    V = _input_transform(x)                                                          # [4,4,n,C,D+2,Ht,Wt]
    dyt = dy.reshape(n, w.shape[0], D, H // 2, 2, W // 2, 2)
    dM = torch.einsum("pi,qj,nozupvq->ijnozuv", AT, AT, dyt)                      # A = (A^T)^T applied on both sides: [4,4,n,Co,D,Ht,Wt]
    dU = torch.stack([torch.einsum("ijnozuv,ijnczuv->ijoc", dM, V[:, :, :, :, k:k + D]) for k in range(3)], dim=2)   # [4,4,3,Co,C]
    dw = torch.einsum("ia,jb,ijkoc->ockab", G, G, dU)
    assert (dw - wr.grad).abs().max().item() < 1e-11


def test_two_dimensional_form_is_the_single_depth_tap_case():
    """ResNet layer3/4 3x3 convolutions: the same transforms with one "depth" tap (kd = 1), images on the batch axis."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 6, 4, 8, generator=g, dtype=torch.float64)
    w = torch.randn(5, 6, 3, 3, generator=g, dtype=torch.float64)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum("ia,jb,ncuvab->ijncuv", BT, BT, d)
    U = torch.einsum("ia,jb,ocab->ijoc", G, G, w)
    Y = torch.einsum("pi,qj,ijoc,ijncuv->noupvq", AT, AT, U, V).reshape(3, 5, 4, 8)
    assert (Y - torch.nn.functional.conv2d(x, w, padding=1)).abs().max().item() < 1e-12
