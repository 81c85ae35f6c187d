"""Per-stage error trace of the engine against the fp32 oracle run on the device (SD-2.1 + SD VAE, 512x512).

For every residual-level stage (each ResnetBlock2D / attention / Transformer2DModel of the VAE encoder, the
UNet's lock-step pass and the VAE decoder) two numbers, relative L2 against the oracle's output of that stage:
  cum    engine running end to end from the pipeline inputs (error accumulated so far);
  local  the engine's stage fed the ORACLE's input of that stage rounded once to the storage dtype (what one
         stage adds on exact input: the floor of a 16-bit-storage pipeline is the root-sum-square of these).
With a third argument "f32s" the engines run with residual_dtype=torch.float32 (fp32 residual stream): `local` is then the
stage on the oracle's UNROUNDED input -- what remains is the 16-bit rounding of the stage's MFMA operands alone.
Usage (GPU box):  python scratch/stage_trace.py [fp16|bf16] [res] [f32s] > gpurun_out/stage_trace_fp16.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import config, weights, unet as U, vae as V            # noqa: E402
from diffews_amd.episodes import make_episode_batch                      # noqa: E402
from diffews_amd.pipeline import MarigoldPipelineRGBLatentNoise          # noqa: E402
from diffews_amd.scheduler import DDIMSchedulerCustomized                # noqa: E402
from oracle import blocks as OB                                          # noqa: E402
from oracle import pipeline as OP                                        # noqa: E402
from oracle.unet import OracleUNet                                       # noqa: E402
from oracle.vae import OracleVAE                                         # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
f32s = len(sys.argv) > 3 and sys.argv[3] == "f32s"
b, nshot = 2, 1
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
kwf = lambda c: {k: v for k, v in c.items() if not k.startswith("_")}
rel = lambda a, r: float((a.float() - r.float()).norm() / (r.float().norm() + 1e-30))
nchw = lambda t: t.permute(0, 3, 1, 2)                 # engine NHWC -> NCHW view
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()

ucfg, vcfg = config.get("sd21_unet"), config.get("sd_vae")
usd = weights.synthetic_unet_state_dict(ucfg, round_to=dt)
vsd = weights.synthetic_vae_state_dict(vcfg, round_to=dt)
te = weights.synthetic_text_embed(ucfg).to(dt).float()
ou = OracleUNet(**kwf(ucfg)); ou.load_state_dict(usd); ou = ou.eval().cuda()
ov = OracleVAE(**kwf(vcfg)); ov.load_state_dict(vsd); ov = ov.eval().cuda()
pipe = MarigoldPipelineRGBLatentNoise(U.MyUNet2DConditionModel(ucfg, usd, torch_dtype=dt), V.AutoencoderKL(vcfg, vsd, torch_dtype=dt),
                                      DDIMSchedulerCustomized(**kwf(config.get("scheduler"))), text_embeds=te.cuda())
if f32s:
    pipe.set_residual_dtype(torch.float32)

# ---- oracle side: record (input, output) of every stage module, in call order
orec = []
def ohook(mod, args, out):
    orec.append((mod, args[0].detach(), out.detach()))
stage_types = (OB.ResnetBlock2D, OB.Transformer2DModel)
for m in list(ov.modules()) + list(ou.modules()):
    if isinstance(m, stage_types) or (isinstance(m, OB.Attention) and m.group_norm is not None):
        m.register_forward_hook(ohook)

# ---- engine side: record the output of every stage object, in call order
erec = []
def wrap(cls):
    orig = cls.__call__
    def rec(self, x, *a, **k):
        y = orig(self, x, *a, **k)
        erec.append((self, y))
        return y
    cls.__call__ = rec
    return orig
originals = {c: wrap(c) for c in (V._VaeResnet, V._VaeAttention, U._Resnet, U._Transformer)}

bt = make_episode_batch(b, nshot, res, seed=41, device="cuda")
with torch.no_grad():
    ref = OP.single_infer(ou, ov, bt["support_imgs"], bt["query_img"], bt["support_masks"], te.cuda())
oracle_calls = list(orec)
r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"], captured=False)
engine_calls = list(erec)
for c, o in originals.items():
    c.__call__ = o

print(f"# stage trace: {str(dt)} storage, residual stream {'fp32' if f32s else str(dt)}, {res}x{res}, b={b}, {nshot}-shot; "
      "relative L2 vs fp32 oracle on device")
n_sup = b * nshot
# oracle call order: encoder x3 (ref imgs, query imgs, masks: P:649-651 order in OP.single_infer), UNet ref pass,
# UNet query pass, decoder.  engine order: ONE encoder pass over [sup ; masks ; query], ONE lock-step UNet, decoder.
n_enc_stage = sum(1 for m in ov.encoder.modules() if isinstance(m, OB.ResnetBlock2D) or (isinstance(m, OB.Attention) and m.group_norm is not None))
n_unet_stage = sum(1 for m in ou.modules() if isinstance(m, stage_types))
n_dec_stage = sum(1 for m in ov.decoder.modules() if isinstance(m, OB.ResnetBlock2D) or (isinstance(m, OB.Attention) and m.group_norm is not None))
enc_o = [oracle_calls[i * n_enc_stage:(i + 1) * n_enc_stage] for i in range(3)]       # z_ref, z_tag, z_gt passes
un_o = oracle_calls[3 * n_enc_stage:3 * n_enc_stage + 2 * n_unet_stage]
dec_o = oracle_calls[3 * n_enc_stage + 2 * n_unet_stage:]
enc_e = engine_calls[:n_enc_stage]
un_e = engine_calls[n_enc_stage:n_enc_stage + n_unet_stage]
dec_e = engine_calls[n_enc_stage + n_unet_stage:]
assert len(dec_o) == n_dec_stage == len(dec_e), (len(dec_o), n_dec_stage, len(dec_e))


def local_err(eng, o_in, o_out, extra=None):
    """engine stage on the oracle's input (rounded once) vs the oracle's output"""
    x = nhwc(o_in).to(torch.float32 if f32s else dt)
    y = originals[type(eng)](eng, x, *(extra or ()))
    return rel(nchw(y), o_out)


# which oracle pass holds which engine batch rows: OP.single_infer encodes (ref, tag, gt) = enc_o[0], [1], [2];
# engine batch = [sup (n_sup) ; masks (n_sup) ; query (b)]
print("\n## VAE encoder (engine batch rows of the QUERY images vs the oracle's z_tag pass)")
print(f"{'stage':34s} {'cum':>10s} {'local':>10s}")
for i, (eng, y) in enumerate(enc_e):
    mod, o_in, o_out = enc_o[1][i]
    name = f"{i:02d} {type(mod).__name__} {tuple(o_out.shape[1:])}"
    print(f"{name:34s} {rel(nchw(y)[2 * n_sup:], o_out):10.3e} {local_err(eng, o_in, o_out):10.3e}")
print(f"{'z_tag (latent mean * 0.18215)':34s} {rel(pipe.encode_rgb(bt['query_img']), ref['z_tag']):10.3e}")

print("\n## UNet, lock-step pass (query rows vs the oracle's query pass; support rows vs its ref pass)")
print(f"{'stage':40s} {'cum query':>10s} {'cum support':>12s}")
for i, (eng, y) in enumerate(un_e):
    mod, _, o_ref = un_o[i]
    _, _, o_q = un_o[n_unet_stage + i]
    name = f"{i:02d} {type(mod).__name__} {tuple(o_q.shape[1:])}"
    print(f"{name:40s} {rel(nchw(y)[n_sup:], o_q):10.3e} {rel(nchw(y)[:n_sup], o_ref):12.3e}")
print(f"{'z0 = -v':40s} {rel(r['z0'], ref['z0']):10.3e}")

print("\n## VAE decoder")
print(f"{'stage':34s} {'cum':>10s} {'local':>10s}")
for i, (eng, y) in enumerate(dec_e):
    mod, o_in, o_out = dec_o[i]
    name = f"{i:02d} {type(mod).__name__} {tuple(o_out.shape[1:])}"
    print(f"{name:34s} {rel(nchw(y), o_out):10.3e} {local_err(eng, o_in, o_out):10.3e}")
seg = (r["dec"].clip(-1, 1) * 0.5 + 0.5) * 255
print(f"decoded mask: mean |diff| = {float((seg - ref['seg'].clip(0, 255)).abs().mean()):.3f} uint8 levels")
