"""DDIM scheduler with the reference's customisation
(/root/reference/marigold/util/scheduler_customized.py:107-180 on top of diffusers DDIMScheduler;
config /root/reference/scheduler_1.0_1.0/scheduler_config.json).

Host-side scalar bookkeeping only; the per-element update is two scalars times tensors.  With the
reference's config (beta_start = beta_end = 1) alphas_cumprod == 0 for every t, so
`pred_original_sample == -model_output`; `z0_is_neg_v()` lets the pipeline fold that sign flip into
the UNet's conv_out epilogue instead of launching anything.
"""
import json
import os
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor


class DDIMSchedulerCustomized:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, power_beta_curve=1.0, **_ignored):
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "scaled_linear_power":  # S:142-144
            betas = torch.linspace(beta_start ** (1 / power_beta_curve), beta_end ** (1 / power_beta_curve),
                                   num_train_timesteps, dtype=torch.float32) ** power_beta_curve
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented")
        if thresholding or rescale_betas_zero_snr:
            raise NotImplementedError("thresholding / rescale_betas_zero_snr are off in the reference config")
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                           steps_offset=steps_offset, prediction_type=prediction_type,
                           clip_sample_range=clip_sample_range, timestep_spacing=timestep_spacing)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        """Accepts both `<path>/<subfolder>/scheduler_config.json` and the reference's
        `./scheduler_1.0_1.0/scheduler_config.json` loaded with subfolder="scheduler" (E:366-367)."""
        cands = [os.path.join(path, subfolder or "", "scheduler_config.json"), os.path.join(path, "scheduler_config.json")]
        for p in cands:
            if os.path.isfile(p):
                with open(p) as f:
                    return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})
        raise FileNotFoundError(f"scheduler_config.json not found under {path}")

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config["num_train_timesteps"]
        if num_inference_steps > T:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        spacing = self.config["timestep_spacing"]
        if spacing == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config["steps_offset"]
        elif spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        elif spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(spacing)
        self.timesteps = torch.from_numpy(ts)  # host tensor: iterating it never syncs the GPU

    def _coeffs(self, timestep):
        t = int(timestep)
        prev_t = t - self.config["num_train_timesteps"] // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def z0_is_neg_v(self, timestep):
        """True when step() reduces to pred_original_sample == -model_output: v-prediction at alpha_bar == 0
        AND no clamp of x0 (with clip_sample the generic step() clamps to +-clip_sample_range; the reference's
        scheduler_config.json sets clip_sample false)."""
        a_t, _ = self._coeffs(timestep)
        return self.config["prediction_type"] == "v_prediction" and a_t == 0.0 and not self.config["clip_sample"]

    def step(self, model_output, timestep, sample, eta=0.0, return_dict=True):
        """DDIM step, eta = 0 (diffusers DDIMScheduler.step formulas (12)/(16))."""
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used on the DiffewS path")
        a_t, a_prev = self._coeffs(timestep)
        b_t = 1.0 - a_t
        pt = self.config["prediction_type"]
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(pt)
        if self.config["clip_sample"]:
            r = self.config["clip_sample_range"]
            x0 = x0.clamp(-r, r)
        prev = a_prev ** 0.5 * x0 + (1.0 - a_prev) ** 0.5 * eps
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)


DDIMScheduler = DDIMSchedulerCustomized
