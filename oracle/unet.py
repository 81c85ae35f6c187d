"""Oracle two-input UNet (test infrastructure; see oracle/__init__.py).

Follows MyUNet2DConditionModel (diffews/models/unet_2d_condition.py):
topology U:185-643, conv_in_ref U:304-306, bank wiring U:645-664,
forward order U:879-1258 (time emb U:991-1015, conv_in|conv_in_ref by
is_target U:1117-1121, down U:1153-1175, mid U:1189-1198, up U:1214-1243,
GN/SiLU/conv_out U:1246-1249).  Parameter names equal the diffusers layout.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import (Downsample2D, ResnetBlock2D, TimestepEmbedding, Transformer2DModel, Upsample2D,
                     timestep_embedding)

SD21_UNET_CONFIG = dict(
    in_channels=4, in_channels_ref=8, out_channels=4,
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    attention_head_dim=(5, 10, 20, 20),  # = number of heads (diffusers naming quirk, U:296-302)
    cross_attention_dim=1024, norm_num_groups=32, norm_eps=1e-5,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    flip_sin_to_cos=True, freq_shift=0, use_linear_projection=True,
)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, heads, cross_dim, groups, eps, n_layers, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(n_layers)])
        if has_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, cout // heads, cout, cross_dim, groups) for _ in range(n_layers)])
        self.has_attn = has_attn
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=1)])
        self.add_down = add_down

    def forward(self, x, temb, ehs):
        outs = ()
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ehs)
            outs += (x,)
        if self.add_down:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, heads, cross_dim, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, c // heads, c, cross_dim, groups)])

    def forward(self, x, temb, ehs):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ehs)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb, heads, cross_dim, groups, eps, n_layers, has_attn, add_up):
        super().__init__()
        resnets = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            resnets.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(resnets)
        if has_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, cout // heads, cout, cross_dim, groups) for _ in range(n_layers)])
        self.has_attn = has_attn
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.add_up = add_up

    def forward(self, x, skips, temb, ehs):
        for i, res in enumerate(self.resnets):
            skip = skips[-1]
            skips = skips[:-1]
            x = res(torch.cat([x, skip], dim=1), temb)
            if self.has_attn:
                x = self.attentions[i](x, ehs)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class OracleUNet(nn.Module):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(SD21_UNET_CONFIG)
        cfg.update(config)
        self.cfg = cfg
        boc = tuple(cfg["block_out_channels"])
        heads = cfg["attention_head_dim"]
        if isinstance(heads, int):
            heads = (heads,) * len(boc)
        groups, eps, cross = cfg["norm_num_groups"], cfg["norm_eps"], cfg["cross_attention_dim"]
        lpb = cfg["layers_per_block"]
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.conv_in_ref = nn.Conv2d(cfg["in_channels_ref"], boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, typ in enumerate(cfg["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(DownBlock(in_c, out_c, temb, heads[i], cross, groups, eps, lpb,
                                              typ == "CrossAttnDownBlock2D", i != len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], temb, heads[-1], cross, groups, eps)
        self.up_blocks = nn.ModuleList()
        rboc, rheads = boc[::-1], tuple(heads)[::-1]
        out_c = rboc[0]
        for i, typ in enumerate(cfg["up_block_types"]):
            prev, out_c = out_c, rboc[i]
            in_c = rboc[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(in_c, out_c, prev, temb, rheads[i], cross, groups, eps, lpb + 1,
                                          typ == "CrossAttnUpBlock2D", i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    # --- bank API (U:645-664) ---
    def _banked(self):
        return [m.attn1 for m in self.modules() if m.__class__.__name__ == "BasicTransformerBlock"]

    def clear_attn_bank(self):
        for a in self._banked():
            a.clear_bank()

    def set_formulation(self, name):
        for a in self._banked():
            a.formulation = name

    def forward(self, sample, timestep, encoder_hidden_states, is_target=True):
        cfg = self.cfg
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.int64, device=sample.device)
        elif timestep.ndim == 0:
            timestep = timestep[None]
        timestep = timestep.to(sample.device).expand(sample.shape[0])   # (device-agnostic: tests may run the oracle on the GPU)
        t_emb = timestep_embedding(timestep, cfg["block_out_channels"][0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        emb = self.time_embedding(t_emb.to(sample.dtype))
        x = self.conv_in(sample) if is_target else self.conv_in_ref(sample)
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            x = blk(x, skips[-n:], emb, encoder_hidden_states)
            skips = skips[:-n]
        x = F.silu(self.conv_norm_out(x))
        return self.conv_out(x)
