import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from forge_amd import convops as co, synthetic as syn
from forge_amd.encoder import Encoder3D
from forge_amd.fusion import affine_act_bwd
dev = torch.device("cuda:0")
enc = Encoder3D(syn.kubric_config())
enc.load_state_dict({k[len("encoder_3d."):]: v for k, v in syn.seeded_state_dict({"encoder_3d." + k: v for k, v in enc.state_dict().items()}, 0).items()})
enc = enc.to(dev).eval()
for p_ in enc.parameters():
    p_.requires_grad_(False)
rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
g = torch.Generator().manual_seed(3)
n, D = 1, 8
D2 = 2 * D
with torch.no_grad():
    enc._heads_hip(torch.randn(n, 128, D, D, D, device=dev), "both")
p = enc._heads_packed_T()
fh = enc.features_head
# piece 1: dgrad of Conv3d(32,16)
gf = torch.randn(n, D2, D2, D2, 16, generator=g).to(dev)
dup = torch.zeros(n, D2, D2, D2, 64, device=dev)
co.narrow_dgrad(gf, p["f3_wT"], dup[..., :32], (n, D2, D2, D2), co.TAPS_3x3x3)
ref = torch.nn.grad.conv3d_input((n, 32, D2, D2, D2), fh[3].weight, gf.permute(0, 4, 1, 2, 3).contiguous(), padding=1).permute(0, 2, 3, 4, 1)
print("narrow_dgrad 32<-16 :", rel(dup[..., :32], ref), " cols16-31 only:", rel(dup[..., 16:32], ref[..., 16:32]), " cols0-15:", rel(dup[..., :16], ref[..., :16]))
print("   interior only:", rel(dup[:, 2:-2, 2:-2, 2:-2, :32], ref[:, 2:-2, 2:-2, 2:-2]))
# piece 2: ConvT dgrad (both heads merged)
gu = torch.randn(n, D2, D2, D2, 64, generator=g).to(dev)
dz = torch.empty(n, D, D, D, 128, device=dev)
co.conv_igemm(gu, 64, 64, None, 0, 0, p["ct_wT"], None, None, None, 1.0, None, None, None, dz, None, (n, D, D, D), (D2, D2, D2), 128, 128, p["ct_taps"], istride=2,
              epilogue=co.EPI_BIAS)
wct = torch.cat([fh[0].weight, enc.density_head[0].weight], dim=1)
ref = F.conv3d(gu.permute(0, 4, 1, 2, 3).contiguous(), wct, stride=2, padding=1).permute(0, 2, 3, 4, 1)       # adjoint of conv_transpose3d(k4,s2,p1)
print("convT dgrad          :", rel(dz, ref), " interior:", rel(dz[:, 1:-1, 1:-1, 1:-1], ref[:, 1:-1, 1:-1, 1:-1]))
# piece 3: affine_act_bwd
y = torch.randn(n, D2, D2, D2, 16, generator=g).to(dev)
sc = torch.rand(16, generator=g).to(dev) + 0.5
o = affine_act_bwd(gf, y, sc, 0.01)
print("affine_act_bwd       :", rel(o, gf * sc * torch.where(y > 0, 1.0, 0.01)))
o = affine_act_bwd(gf, gf, sc, 1.0)
print("affine_act_bwd slope1:", rel(o, gf * sc))
