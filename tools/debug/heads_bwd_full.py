import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from forge_amd import convops as co, synthetic as syn
from forge_amd.encoder import Encoder3D, _HeadsFrozen
dev = torch.device("cuda:0")
enc = Encoder3D(syn.kubric_config())
enc.load_state_dict({k[len("encoder_3d."):]: v for k, v in syn.seeded_state_dict({"encoder_3d." + k: v for k, v in enc.state_dict().items()}, 0).items()})
enc = enc.to(dev).eval()
for p_ in enc.parameters():
    p_.requires_grad_(False)
rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
g = torch.Generator().manual_seed(3)
DD = int(os.environ.get("DD", "8"))
z0 = torch.randn(1, 128, DD, DD, DD, generator=g).to(dev)
# stock torch modules (MIOpen) as the reference
zs = z0.clone().requires_grad_(True)
fs = enc.features_head(zs)
wf = torch.linspace(-1, 1, fs.numel(), device=dev).reshape(fs.shape)
(fs * wf).sum().backward()
zf = z0.clone().requires_grad_(True)
feat, dens = _HeadsFrozen.apply(zf, enc)
print("forward feat", rel(feat, fs))
(feat * wf).sum().backward()
print("dz frozen vs stock", rel(zf.grad, zs.grad))
# stage by stage with torch
fh = enc.features_head
up_ref = fh[2](fh[1](fh[0](z0)))                      # [1,32,16,16,16]
up_ref.requires_grad_(True)
f2 = fh[4](fh[3](up_ref))
(f2 * wf).sum().backward()
dup_ref = up_ref.grad.permute(0, 2, 3, 4, 1)
p = enc._heads_packed_T()
from forge_amd.fusion import affine_act_bwd
rows = wf.permute(0, 2, 3, 4, 1).contiguous()
gf = affine_act_bwd(rows, rows, p["f4_scale"], 1.0)
dup = torch.zeros(1, 2 * DD, 2 * DD, 2 * DD, 64, device=dev)
co.narrow_dgrad(gf, p["f3_wT"], dup[..., :32], (1, 2 * DD, 2 * DD, 2 * DD), co.TAPS_3x3x3)
print("dup stage", rel(dup[..., :32], dup_ref))
print("f4 scale vs module", rel(p["f4_scale"], fh[4].weight / torch.sqrt(fh[4].running_var + fh[4].eps)))
# ---- stage: gu (gradient wrt the transposed-conv pre-activation) and dz
pre = fh[0](z0).detach().requires_grad_(True)
f3 = fh[4](fh[3](fh[2](fh[1](pre))))
(f3 * wf).sum().backward()
gu_ref = pre.grad.permute(0, 2, 3, 4, 1)
with torch.no_grad():
    _, _, up, d8 = enc._heads_hip(z0, "both", keep=True)
gu = affine_act_bwd(dup, up, p["ct_scale"], 0.01)
print("gu stage", rel(gu[..., :32], gu_ref), " up vs ref", rel(up[..., :32], fh[2](fh[1](pre)).permute(0, 2, 3, 4, 1)))
dz = torch.empty(1, DD, DD, DD, 128, device=dev)
co.conv_igemm(gu, 64, 64, None, 0, 0, p["ct_wT"], None, None, None, 1.0, None, None, None, dz, None, (1, DD, DD, DD), (2 * DD, 2 * DD, 2 * DD), 128, 128,
              p["ct_taps"], istride=2, epilogue=co.EPI_BIAS)
print("dz stage (script)", rel(dz.permute(0, 4, 1, 2, 3), zs.grad), " dz real backward", rel(zf.grad, zs.grad))
gu2 = gu.clone(); gu2[..., 32:] = 0
co.conv_igemm(gu2, 64, 64, None, 0, 0, p["ct_wT"], None, None, None, 1.0, None, None, None, dz, None, (1, DD, DD, DD), (2 * DD, 2 * DD, 2 * DD), 128, 128,
              p["ct_taps"], istride=2, epilogue=co.EPI_BIAS)
print("dz with dens half zeroed", rel(dz.permute(0, 4, 1, 2, 3), zs.grad), " max|gu dens half|", gu[..., 32:].abs().max().item(), " max|dup dens half|", dup[..., 32:].abs().max().item())
