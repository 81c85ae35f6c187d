import sys, torch
sys.path.insert(0, '/root/repo')
from diffews_amd import weights, config
from diffews_amd.unet import MyUNet2DConditionModel
from oracle.unet import OracleUNet
def rel(a,b): a,b=a.float().cpu(),b.float().cpu(); return float((a-b).norm()/(b.norm()+1e-30))
for dt in (torch.float16, torch.bfloat16):
    ucfg = config.get("tiny_unet")
    usd = weights.synthetic_unet_state_dict(ucfg, round_to=dt)
    te = weights.synthetic_text_embed(ucfg).to(dt).float()
    unet = MyUNet2DConditionModel(ucfg, usd, torch_dtype=dt)
    ou = OracleUNet(**{k: v for k, v in ucfg.items() if not k.startswith("_")}); ou.load_state_dict(usd); ou.eval()
    g = torch.Generator().manual_seed(9)
    b, s = 2, 2
    zr = (torch.randn(b * s, 8, 16, 16, generator=g) * 0.5).cuda(); zq = (torch.randn(b, 4, 16, 16, generator=g) * 0.5).cuda()
    ehs, ehs_r = te.repeat(b, 1, 1).cuda(), te.repeat(b * s, 1, 1).cuda()
    plain = unet.forward_pair(zr, zq, 1, ehs_r, ehs)
    unet.fold_conditioning(1, te)
    fold = unet.forward_pair(zr, zq, 1)
    with torch.no_grad():
        ou.clear_attn_bank(); ou(zr.cpu(), 1, ehs_r.cpu(), is_target=False); ref = ou(zq.cpu(), 1, ehs.cpu()); ou.clear_attn_bank()
    print(dt, "fold vs plain", rel(fold, plain), "plain vs ref", rel(plain, ref), "fold vs ref", rel(fold, ref))
