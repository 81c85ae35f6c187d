"""Other resolutions (ragged tile cases): engine vs fp32 oracle on the device at 384^2, 448^2, 768^2, 320x512."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import config, weights
from diffews_amd.pipeline import MarigoldPipelineRGBLatentNoise
from diffews_amd.scheduler import DDIMSchedulerCustomized
from diffews_amd.unet import MyUNet2DConditionModel
from diffews_amd.vae import AutoencoderKL
from oracle import pipeline as OP
from oracle.unet import OracleUNet
from oracle.vae import OracleVAE
rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
torch.backends.cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
kwf = lambda c: {k: v for k, v in c.items() if not k.startswith("_")}
dt = torch.float16
ucfg, vcfg = config.get("sd21_unet"), config.get("sd_vae")
usd = weights.synthetic_unet_state_dict(ucfg, round_to=dt); vsd = weights.synthetic_vae_state_dict(vcfg, round_to=dt)
te = weights.synthetic_text_embed(ucfg).to(dt).float()
ou = OracleUNet(**kwf(ucfg)); ou.load_state_dict(usd); ou = ou.eval().cuda()
ov = OracleVAE(**kwf(vcfg)); ov.load_state_dict(vsd); ov = ov.eval().cuda()
pipe = MarigoldPipelineRGBLatentNoise(MyUNet2DConditionModel(ucfg, usd, torch_dtype=dt), AutoencoderKL(vcfg, vsd, torch_dtype=dt),
                                      DDIMSchedulerCustomized(**kwf(config.get("scheduler"))), text_embeds=te.cuda())
g = torch.Generator().manual_seed(5)
for (H, W, b, s) in [(384, 384, 1, 1), (448, 448, 2, 1), (768, 768, 1, 1), (320, 512, 1, 2), (64, 64, 3, 1)]:
    sup = (torch.rand(b * s, 3, H, W, generator=g) * 2 - 1).cuda(); qry = (torch.rand(b, 3, H, W, generator=g) * 2 - 1).cuda()
    m = torch.zeros(b * s, 1, H, W); m[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1
    msk = (m.repeat(1, 3, 1, 1) * 2 - 1).cuda()
    with torch.no_grad():
        ref = OP.single_infer(ou, ov, sup, qry, msk, te.cuda())
    r = pipe.run_episodes(sup, qry, msk)
    print(f"{H}x{W} b={b} s={s}: z0 rel {rel(r['z0'], ref['z0']):.3e}  finite={bool(torch.isfinite(r['z0']).all())}", flush=True)
