"""Model configurations for the hot path.

The reference fixes only the UNet topology defaults
(/root/reference/diffews/models/unet_2d_condition.py:185-241) plus
`in_channels_ref=8` (U:189); the SD-2.1 hyper-parameters below come from the
SD-2.1 `unet/config.json` / `vae/config.json` a DiffewS checkpoint ships with
(SURVEY.md section 2.2).  `tiny_*` are small-width variants used by the parity
tests so that the CPU oracle finishes in seconds; they keep head_dim == 64 and
every channel count a multiple of 64, the same constraints the HIP kernels
assume for SD-2.1.
"""
import copy

SD21_UNET = dict(
    _class_name="UNet2DConditionModel",
    in_channels=4, in_channels_ref=8, out_channels=4,
    block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
    attention_head_dim=[5, 10, 20, 20],  # number of heads per level (diffusers naming quirk, U:296-302)
    cross_attention_dim=1024, norm_num_groups=32, norm_eps=1e-5,
    down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    flip_sin_to_cos=True, freq_shift=0, use_linear_projection=True, sample_size=96,
)

SD_VAE = dict(
    _class_name="AutoencoderKL",
    in_channels=3, out_channels=3, latent_channels=4,
    block_out_channels=[128, 256, 512, 512], layers_per_block=2, norm_num_groups=32,
    scaling_factor=0.18215,
)

TINY_UNET = dict(SD21_UNET, block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4],
                 cross_attention_dim=128, sample_size=16)
TINY_VAE = dict(SD_VAE, block_out_channels=[64, 128, 256, 256])

# degenerate DDIM of the reference: /root/reference/scheduler_1.0_1.0/scheduler_config.json
SCHEDULER = dict(
    _class_name="DDIMScheduler", beta_start=1.0, beta_end=1.0, beta_schedule="scaled_linear",
    clip_sample=False, num_train_timesteps=1000, prediction_type="v_prediction",
    set_alpha_to_one=False, steps_offset=1, timestep_spacing="leading", trained_betas=None,
)


def get(name):
    return copy.deepcopy({"sd21_unet": SD21_UNET, "sd_vae": SD_VAE, "tiny_unet": TINY_UNET,
                          "tiny_vae": TINY_VAE, "scheduler": SCHEDULER}[name])
