"""fp32 residual stream (residual_dtype=torch.float32) vs the 16-bit stream: z0 / z_tag error against the fp32 oracle on the
device at the exact BASELINE shapes, and the step time of both modes (eager, batch 4 x 512^2 1-shot)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import config, weights
from diffews_amd.episodes import make_episode_batch
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
dts = [torch.float16, torch.bfloat16] if len(sys.argv) < 2 else [getattr(torch, sys.argv[1])]
for dt in dts:
    ucfg, vcfg = config.get("sd21_unet"), config.get("sd_vae")
    usd = weights.synthetic_unet_state_dict(ucfg, round_to=dt); vsd = weights.synthetic_vae_state_dict(vcfg, round_to=dt)
    te = weights.synthetic_text_embed(ucfg).to(dt).float()
    ou = OracleUNet(**kwf(ucfg)); ou.load_state_dict(usd); ou = ou.eval().cuda()
    ov = OracleVAE(**kwf(vcfg)); ov.load_state_dict(vsd); ov = ov.eval().cuda()
    refs = {}
    for b, nshot, res in ((4, 1, 512), (2, 5, 512), (1, 1, 256)):
        bt = make_episode_batch(b, nshot, res, seed=40 + nshot + b, device="cuda")
        with torch.no_grad():
            ref = OP.single_infer(ou, ov, bt["support_imgs"], bt["query_img"], bt["support_masks"], te.cuda())
        refs[(b, nshot, res)] = (bt, {k: ref[k].clone() for k in ("z0", "z_tag", "seg")})
        del ref
    del ou, ov
    torch.cuda.empty_cache()
    for modes in (((None, None), (torch.float32, torch.float32)) if os.environ.get("F32_ONLY") else ((None, None), (torch.float32, None), (None, torch.float32), (torch.float32, torch.float32))):
        rv, ru = modes
        pipe = MarigoldPipelineRGBLatentNoise(MyUNet2DConditionModel(ucfg, usd, torch_dtype=dt, residual_dtype=ru),
                                              AutoencoderKL(vcfg, vsd, torch_dtype=dt, residual_dtype=rv),
                                              DDIMSchedulerCustomized(**kwf(config.get("scheduler"))), text_embeds=te.cuda())
        for key, (bt, ref) in refs.items():
            r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"])
            seg = (r["dec"].clip(-1, 1) * 0.5 + 0.5) * 255
            print(f"{str(dt):15s} vae_stream={str(rv):14s} unet_stream={str(ru):14s} b={key[0]} {key[1]}-shot {key[2]}^2: z0 rel {rel(r['z0'], ref['z0']):.3e}  "
                  f"z_tag rel {rel(pipe.encode_rgb(bt['query_img']), ref['z_tag']):.3e}  seg mean|d| {float((seg - ref['seg'].clip(0,255)).abs().mean()):.3f}", flush=True)
        bt = refs[(4, 1, 512)][0]
        args = (bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"])
        for _ in range(2):
            pipe.run_episodes(*args, captured=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pipe.run_episodes(*args, captured=True)
        torch.cuda.synchronize()
        print(f"   step (b=4, 512^2, 1-shot, graph): {(time.perf_counter() - t0) * 100:.2f} ms", flush=True)
        del pipe
        torch.cuda.empty_cache()
