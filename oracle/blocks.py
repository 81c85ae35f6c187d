"""Oracle building blocks (test infrastructure; see oracle/__init__.py).

Restates the diffusers-0.25.0 modules the reference UNet/VAE are assembled from
(get_down_block / get_up_block / UNetMidBlock2DCrossAttn at
diffews/models/unet_2d_condition.py:473,503,584) with the *same parameter
names*, so ``state_dict()`` keys equal the diffusers checkpoint layout.
Everything is NCHW fp32 and uses only torch.nn.functional primitives.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers `Timesteps` (unet_2d_condition.py:320): sinusoidal, fp32."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """GN-SiLU-conv3x3 (+temb) GN-SiLU-conv3x3, 1x1 shortcut iff Cin != Cout."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    """conv3x3 stride 2; padding=1 (UNet) or F.pad(0,1,0,1)+padding 0 (VAE encoder)."""

    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    """diffusers `Attention` + the reference's K/V bank (MyAttention,
    diffews/models/attention_processor.py:39-50).

    ``formulation`` selects which of the reference's three processors is
    restated: 'xformers' (A:182-288, handles n-shot by folding ref-batch into
    tokens), 'sdpa' (A:291-383) or 'vanilla' (A:104-180, baddbmm+softmax+bmm);
    the latter two concatenate the bank without the reshape (1-shot only).
    """

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False,
                 norm_num_groups=None, eps=1e-5, residual_connection=False, has_bank=False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps) if norm_num_groups else None
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.has_bank = has_bank
        self.k_bank = None
        self.v_bank = None
        self.formulation = "xformers"

    def clear_bank(self):
        self.k_bank = None
        self.v_bank = None

    def head_to_batch_dim(self, t):
        b, n, _ = t.shape
        return t.reshape(b, n, self.heads, self.dim_head).permute(0, 2, 1, 3).reshape(b * self.heads, n, self.dim_head)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, n, d).permute(0, 2, 1, 3).reshape(b, n, self.heads * d)

    def _bank_tokens(self, bank, batch_size):
        if self.formulation == "xformers":
            # A:256-257: [B_ref*h, N, d] -> [B_ref, N, C] -> [b, s*N, C] -> [b*h, s*N, d]
            t = self.batch_to_head_dim(bank)
            return self.head_to_batch_dim(t.reshape(batch_size, -1, t.shape[-1]))
        return bank  # A:156 / A:354: plain concat, valid only when B_ref == b

    def forward(self, hidden_states, encoder_hidden_states=None):
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        batch_size = hidden_states.shape[0]
        if self.group_norm is not None:
            hidden_states = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        q = self.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        q, k, v = (self.head_to_batch_dim(t) for t in (q, k, v))
        if self.has_bank:
            if self.k_bank is None:
                self.k_bank, self.v_bank = k, v
            else:
                k = torch.cat([k, self._bank_tokens(self.k_bank, batch_size)], dim=1)
                v = torch.cat([v, self._bank_tokens(self.v_bank, batch_size)], dim=1)
        if self.formulation == "vanilla":
            scores = torch.baddbmm(
                torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                q, k.transpose(-1, -2), beta=0, alpha=self.scale)
            out = torch.bmm(scores.softmax(dim=-1), v)
        else:
            out = F.scaled_dot_product_attention(q[None], k[None], v[None], scale=self.scale)[0]
        out = self.batch_to_head_dim(out)
        out = self.to_out[0](out)
        if input_ndim == 4:
            out = out.transpose(-1, -2).reshape(b, c, hh, ww)
        if self.residual_connection:
            out = out + residual
        return out


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """pre-LN attn1 (self, banked) -> attn2 (cross) -> GEGLU FF, each residual."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head, has_bank=True)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    """use_linear_projection=True form: GN(1e-6) -> tokens -> Linear -> blocks -> Linear -> +res."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32, num_layers=1):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states):
        b, c, h, w = x.shape
        residual = x
        t = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        t = self.proj_in(t)
        for blk in self.transformer_blocks:
            t = blk(t, encoder_hidden_states)
        t = self.proj_out(t)
        return t.reshape(b, h, w, c).permute(0, 3, 1, 2) + residual
