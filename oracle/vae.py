"""Oracle AutoencoderKL encoder/decoder (test infrastructure; see oracle/__init__.py).

Restates diffusers-0.25.0 `AutoencoderKL` as the reference pipeline drives it:
`vae.encoder` + `vae.quant_conv` (marigold_pipeline_rgb_latent_noise.py:852-853)
and `vae.post_quant_conv` + `vae.decoder` (P:901-902).  Parameter names equal
the diffusers checkpoint layout.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import Attention, Downsample2D, ResnetBlock2D, Upsample2D

SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4,
                     block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32)


class VaeMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, 1e-6) for _ in range(2)])
        # attention_head_dim = c  => one head of dim c; GN + biased q/k/v + residual
        self.attentions = nn.ModuleList([
            Attention(c, None, heads=1, dim_head=c, bias=True, norm_num_groups=groups, eps=1e-6,
                      residual_connection=True)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class EncBlock(nn.Module):
    def __init__(self, cin, cout, groups, n_layers, add_down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(n_layers)])
        self.add_down = add_down
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.add_down:
            x = self.downsamplers[0](x)
        return x


class DecBlock(nn.Module):
    def __init__(self, cin, cout, groups, n_layers, add_up):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(n_layers)])
        self.add_up = add_up
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g, lpb = tuple(cfg["block_out_channels"]), cfg["norm_num_groups"], cfg["layers_per_block"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(EncBlock(in_c, out_c, g, lpb, i != len(boc) - 1))
        self.mid_block = VaeMid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g, lpb = tuple(cfg["block_out_channels"]), cfg["norm_num_groups"], cfg["layers_per_block"]
        rboc = boc[::-1]
        self.conv_in = nn.Conv2d(cfg["latent_channels"], rboc[0], 3, padding=1)
        self.mid_block = VaeMid(rboc[0], g)
        self.up_blocks = nn.ModuleList()
        out_c = rboc[0]
        for i in range(len(boc)):
            prev, out_c = out_c, rboc[i]
            self.up_blocks.append(DecBlock(prev, out_c, g, lpb + 1, i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class OracleVAE(nn.Module):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(SD_VAE_CONFIG)
        cfg.update(config)
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        lc = cfg["latent_channels"]
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)

    def encode_mean(self, x):
        """P:852-859: mean half of quant_conv(encoder(x)); no sampling at inference."""
        moments = self.quant_conv(self.encoder(x))
        mean, _logvar = torch.chunk(moments, 2, dim=1)
        return mean

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
