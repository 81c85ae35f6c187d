"""What plain PyTorch-ROCm (MIOpen / rocBLAS / SDPA, eager) makes of the same graph on the same MI355X:
the oracle modules moved to the device in bf16 -- the closest stand-in for "the reference's diffusers path on
this GPU" available offline.  Context for bench.py's number, not a baseline the bench reports."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import config, weights
from diffews_amd.episodes import make_episode_batch
from oracle import pipeline as OP
from oracle.unet import OracleUNet
from oracle.vae import OracleVAE
kwf = lambda c: {k: v for k, v in c.items() if not k.startswith("_")}
ucfg, vcfg = config.get("sd21_unet"), config.get("sd_vae")
for dt in (torch.bfloat16, torch.float16):
    ou = OracleUNet(**kwf(ucfg)); ou.load_state_dict(weights.synthetic_unet_state_dict(ucfg)); ou = ou.eval().cuda().to(dt)
    ov = OracleVAE(**kwf(vcfg)); ov.load_state_dict(weights.synthetic_vae_state_dict(vcfg)); ov = ov.eval().cuda().to(dt)
    te = weights.synthetic_text_embed(ucfg).cuda().to(dt)
    b = 4
    bt = make_episode_batch(b, 1, 512, seed=1, device="cuda")
    args = [bt[k].to(dt) for k in ("support_imgs", "query_img", "support_masks")]
    with torch.no_grad():
        for _ in range(2):
            OP.single_infer(ou, ov, *args, te)
        torch.cuda.synchronize(); t0 = time.time()
        n = 5
        for _ in range(n):
            OP.single_infer(ou, ov, *args, te)
        torch.cuda.synchronize(); dtm = (time.time() - t0) / n
    print(f"torch eager {str(dt):15s}: {dtm*1e3:8.1f} ms per {b}-episode step = {b/dtm:6.2f} episodes/s", flush=True)
    del ou, ov
    torch.cuda.empty_cache()
