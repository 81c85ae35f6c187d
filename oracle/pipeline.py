"""Oracle episode pipeline + scheduler (test infrastructure; see oracle/__init__.py).

Restates MarigoldPipelineRGBLatentNoise.single_infer / __call__ for mode='seg'
(diffews/marigold_pipeline_rgb_latent_noise.py:223-583, 616-836, 839-905) and
DDIMSchedulerCustomized + diffusers DDIMScheduler.step
(marigold/util/scheduler_customized.py:107-180, scheduler_1.0_1.0/scheduler_config.json).
"""
import json

import numpy as np
import torch
import torch.nn.functional as F

LATENT_SCALE = 0.18215  # P:120-124


class OracleDDIM:
    """DDIM with the reference's config: beta==1 => alphas_cumprod==0 (S:128-152)."""

    def __init__(self, num_train_timesteps=1000, beta_start=1.0, beta_end=1.0, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", set_alpha_to_one=False, steps_offset=1,
                 timestep_spacing="leading", **_unused):
        assert beta_schedule == "scaled_linear" and timestep_spacing == "leading"
        self.T = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.prediction_type = prediction_type
        self.steps_offset = steps_offset
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_json(cls, path):
        with open(path) as f:
            return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})

    def set_timesteps(self, n):
        self.num_inference_steps = n
        step_ratio = self.T // n
        ts = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample):
        t = int(timestep)
        prev_t = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        assert self.prediction_type == "v_prediction"
        pred_x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        pred_eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        prev_sample = a_prev ** 0.5 * pred_x0 + (1 - a_prev) ** 0.5 * pred_eps  # eta = 0
        return prev_sample, pred_x0


@torch.no_grad()
def single_infer(unet, vae, rgb_in_ref, rgb_in_tag, gt_in_ref, text_embed, test_timestep=1,
                 num_inference_steps=1, scheduler=None):
    """P:617-802 for mode='seg'.  Returns dict with every intermediate the parity tests compare."""
    sched = scheduler or OracleDDIM()
    sched.set_timesteps(num_inference_steps)
    enc = lambda x: vae.encode_mean(x) * LATENT_SCALE  # P:839-862
    z_ref, z_tag, z_gt = enc(rgb_in_ref), enc(rgb_in_tag), enc(gt_in_ref)
    cond_ref = torch.cat([z_ref, z_gt], dim=1)  # P:674
    z = z_tag.clone()  # P:675
    b = z_tag.shape[0]
    ehs = text_embed.repeat(b, 1, 1)  # P:690
    ehs_ref = ehs.repeat(z_ref.shape[0] // b, 1, 1)  # P:692
    noise_pred = None
    for t in sched.timesteps:
        unet.clear_attn_bank()  # P:715
        unet(cond_ref, t * test_timestep, ehs_ref, is_target=False)  # P:719-720
        noise_pred = unet(z, t * test_timestep, ehs)  # P:721-723
        unet.clear_attn_bank()  # P:725
        z, z0 = sched.step(noise_pred, t, z)  # P:764-765
    seg = vae.decode(z0 / LATENT_SCALE).clip(-1, 1)  # P:899-903
    seg = (seg.clip(-1.0, 1.0) * 0.5 + 0.5) * 255  # P:790-795
    return dict(z_ref=z_ref, z_tag=z_tag, z_gt=z_gt, noise_pred=noise_pred, z0=z0, seg=seg)


@torch.no_grad()
def pipeline_call(unet, vae, input_images, text_embed, test_timestep=1, denoising_steps=1):
    """P:223-545 for tensor inputs, ensemble_size=1, mode='seg': uint8 HWC masks, one per query."""
    sup, qry, msk = input_images
    for t in input_images:
        assert t.min() >= -1.0 and t.max() <= 1.0  # P:309
    out = single_infer(unet, vae, sup, qry, msk, text_embed, test_timestep, denoising_steps)
    pred = out["seg"]
    pred = F.interpolate(pred, qry.shape[-2:], mode="nearest")  # P:474
    u8 = pred.clip(0, 255).cpu().numpy().astype(np.uint8)  # P:534
    return [np.moveaxis(m, 0, -1) for m in u8], out  # chw2hwc, P:538


def threshold_mask(seg_u8_hwc, r_threshold=0.25, threshold=0.0):
    """evaluation_util/main_oss.py:128-137: to_tensor (/255), mean over channels, dynamic threshold."""
    pred = torch.from_numpy(np.moveaxis(seg_u8_hwc, -1, 0)).float().div(255)[None]
    if r_threshold > 0:
        pred = (pred.mean(dim=1) > pred.max() * r_threshold).float()
    if threshold > 0:
        pred = (pred.mean(dim=1) > threshold).float()
    return pred
