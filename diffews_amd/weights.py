"""Checkpoint layout: parameter names/shapes in the diffusers layout, synthetic
weights, and safetensors I/O.

The reference loads `unet/config.json` + `unet/diffusion_pytorch_model.safetensors`
with diffusers parameter names plus `conv_in_ref.{weight,bias}`
(/root/reference/evaluation_util/main_oss.py:338-345,
 /root/reference/diffews/models/unet_2d_condition.py:304-306); `vae/` likewise
(E:347-349).  This module is the single description of that layout for the
MI355X engine.  No weights ship with the reference, so benchmarks and tests use
`synthetic_*` (seeded, identical on CPU oracle and GPU engine).
"""
import json
import os
from collections import OrderedDict

import torch

from . import config as _config


def _resnet(out, p, cin, cout, temb):
    out[p + "norm1.weight"] = (cin,)
    out[p + "norm1.bias"] = (cin,)
    out[p + "conv1.weight"] = (cout, cin, 3, 3)
    out[p + "conv1.bias"] = (cout,)
    if temb:
        out[p + "time_emb_proj.weight"] = (cout, temb)
        out[p + "time_emb_proj.bias"] = (cout,)
    out[p + "norm2.weight"] = (cout,)
    out[p + "norm2.bias"] = (cout,)
    out[p + "conv2.weight"] = (cout, cout, 3, 3)
    out[p + "conv2.bias"] = (cout,)
    if cin != cout:
        out[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + "conv_shortcut.bias"] = (cout,)


def _transformer(out, p, c, cross):
    out[p + "norm.weight"] = (c,)
    out[p + "norm.bias"] = (c,)
    out[p + "proj_in.weight"] = (c, c)
    out[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        out[b + n + ".weight"] = (c,)
        out[b + n + ".bias"] = (c,)
    for a, kv in (("attn1", c), ("attn2", cross)):
        out[b + a + ".to_q.weight"] = (c, c)
        out[b + a + ".to_k.weight"] = (c, kv)
        out[b + a + ".to_v.weight"] = (c, kv)
        out[b + a + ".to_out.0.weight"] = (c, c)
        out[b + a + ".to_out.0.bias"] = (c,)
    out[b + "ff.net.0.proj.weight"] = (8 * c, c)
    out[b + "ff.net.0.proj.bias"] = (8 * c,)
    out[b + "ff.net.2.weight"] = (c, 4 * c)
    out[b + "ff.net.2.bias"] = (c,)
    out[p + "proj_out.weight"] = (c, c)
    out[p + "proj_out.bias"] = (c,)


def unet_param_shapes(cfg):
    """name -> shape for MyUNet2DConditionModel (U:185-643), diffusers key layout."""
    boc = list(cfg["block_out_channels"])
    cross, lpb = cfg["cross_attention_dim"], cfg["layers_per_block"]
    temb = boc[0] * 4
    out = OrderedDict()
    out["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    out["conv_in.bias"] = (boc[0],)
    out["conv_in_ref.weight"] = (boc[0], cfg["in_channels_ref"], 3, 3)
    out["conv_in_ref.bias"] = (boc[0],)
    out["time_embedding.linear_1.weight"] = (temb, boc[0])
    out["time_embedding.linear_1.bias"] = (temb,)
    out["time_embedding.linear_2.weight"] = (temb, temb)
    out["time_embedding.linear_2.bias"] = (temb,)
    oc = boc[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        ic, oc = oc, boc[i]
        for j in range(lpb):
            _resnet(out, f"down_blocks.{i}.resnets.{j}.", ic if j == 0 else oc, oc, temb)
            if typ == "CrossAttnDownBlock2D":
                _transformer(out, f"down_blocks.{i}.attentions.{j}.", oc, cross)
        if i != len(boc) - 1:
            out[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            out[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (oc,)
    c = boc[-1]
    _resnet(out, "mid_block.resnets.0.", c, c, temb)
    _transformer(out, "mid_block.attentions.0.", c, cross)
    _resnet(out, "mid_block.resnets.1.", c, c, temb)
    rboc = boc[::-1]
    oc = rboc[0]
    for i, typ in enumerate(cfg["up_block_types"]):
        prev, oc = oc, rboc[i]
        ic = rboc[min(i + 1, len(boc) - 1)]
        for j in range(lpb + 1):
            skip = ic if j == lpb else oc
            rin = prev if j == 0 else oc
            _resnet(out, f"up_blocks.{i}.resnets.{j}.", rin + skip, oc, temb)
            if typ == "CrossAttnUpBlock2D":
                _transformer(out, f"up_blocks.{i}.attentions.{j}.", oc, cross)
        if i != len(boc) - 1:
            out[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            out[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (oc,)
    out["conv_norm_out.weight"] = (boc[0],)
    out["conv_norm_out.bias"] = (boc[0],)
    out["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
    out["conv_out.bias"] = (cfg["out_channels"],)
    return out


def _vae_mid(out, p, c):
    _resnet(out, p + "resnets.0.", c, c, 0)
    a = p + "attentions.0."
    out[a + "group_norm.weight"] = (c,)
    out[a + "group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[a + n + ".weight"] = (c, c)
        out[a + n + ".bias"] = (c,)
    _resnet(out, p + "resnets.1.", c, c, 0)


def vae_param_shapes(cfg):
    """name -> shape for diffusers AutoencoderKL as driven by P:852-853, 901-902."""
    boc, lpb, lc = list(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["latent_channels"]
    out = OrderedDict()
    out["encoder.conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    out["encoder.conv_in.bias"] = (boc[0],)
    oc = boc[0]
    for i in range(len(boc)):
        ic, oc = oc, boc[i]
        for j in range(lpb):
            _resnet(out, f"encoder.down_blocks.{i}.resnets.{j}.", ic if j == 0 else oc, oc, 0)
        if i != len(boc) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (oc,)
    _vae_mid(out, "encoder.mid_block.", boc[-1])
    out["encoder.conv_norm_out.weight"] = (boc[-1],)
    out["encoder.conv_norm_out.bias"] = (boc[-1],)
    out["encoder.conv_out.weight"] = (2 * lc, boc[-1], 3, 3)
    out["encoder.conv_out.bias"] = (2 * lc,)
    out["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    out["quant_conv.bias"] = (2 * lc,)
    out["post_quant_conv.weight"] = (lc, lc, 1, 1)
    out["post_quant_conv.bias"] = (lc,)
    rboc = boc[::-1]
    out["decoder.conv_in.weight"] = (rboc[0], lc, 3, 3)
    out["decoder.conv_in.bias"] = (rboc[0],)
    _vae_mid(out, "decoder.mid_block.", rboc[0])
    oc = rboc[0]
    for i in range(len(boc)):
        prev, oc = oc, rboc[i]
        for j in range(lpb + 1):
            _resnet(out, f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else oc, oc, 0)
        if i != len(boc) - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (oc,)
    out["decoder.conv_norm_out.weight"] = (boc[0],)
    out["decoder.conv_norm_out.bias"] = (boc[0],)
    out["decoder.conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
    out["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return out


def _synthetic(shapes, seed, round_to=None):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        is_norm = ".norm" in name or name.startswith("norm") or "group_norm" in name or "conv_norm_out" in name
        if is_norm:
            t = torch.randn(shape, generator=g) * 0.1
            if leaf == "weight":
                t = t + 1.0
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.02
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (fan_in ** -0.5)
        if round_to is not None:  # weights representable exactly in the engine's storage dtype
            t = t.to(round_to).float()
        sd[name] = t
    return sd


def synthetic_unet_state_dict(cfg, seed=1234, round_to=None):
    """Seeded synthetic UNet weights; conv_in_ref follows the reference's weight surgery
    (train_tools/load_ckpt_and_modify_ref8in_tag4in.py:21-24): repeat(conv_in, 2 along Cin)/2."""
    sd = _synthetic(unet_param_shapes(cfg), seed, round_to)
    if cfg["in_channels_ref"] == 2 * cfg["in_channels"]:
        w = sd["conv_in.weight"].repeat(1, 2, 1, 1) / 2
        sd["conv_in_ref.weight"] = w if round_to is None else w.to(round_to).float()
        sd["conv_in_ref.bias"] = sd["conv_in.bias"].clone()
    return sd


def synthetic_vae_state_dict(cfg, seed=4321, round_to=None):
    return _synthetic(vae_param_shapes(cfg), seed, round_to)


def synthetic_text_embed(cfg, seed=3, tokens=2):
    """Stand-in for CLIP("") -> [1, 2, cross_attention_dim] (P:590-601; constant per checkpoint)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, tokens, cfg["cross_attention_dim"], generator=g)


# ---------------------------------------------------------------- diffusers directory I/O

def _model_dir(path, subfolder):
    return os.path.join(path, subfolder) if subfolder else path


def load_config(path, subfolder=None, filename="config.json"):
    d = _model_dir(path, subfolder)
    p = os.path.join(d, filename)
    if not os.path.isfile(p) and subfolder:  # ./scheduler_1.0_1.0 keeps the JSON at top level (E:367)
        p = os.path.join(path, filename)
    with open(p) as f:
        return json.load(f)


def load_state_dict(path, subfolder=None):
    from safetensors.torch import load_file
    d = _model_dir(path, subfolder)
    for fn in ("diffusion_pytorch_model.safetensors", "model.safetensors"):
        p = os.path.join(d, fn)
        if os.path.isfile(p):
            return load_file(p)
    p = os.path.join(d, "diffusion_pytorch_model.bin")
    if os.path.isfile(p):
        return torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no diffusers weights under {d}")


def save_pretrained(path, cfg, state_dict, subfolder=None):
    from safetensors.torch import save_file
    d = _model_dir(path, subfolder)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    save_file({k: v.contiguous() for k, v in state_dict.items()},
              os.path.join(d, "diffusion_pytorch_model.safetensors"))


def check_state_dict(state_dict, shapes, what):
    missing = [k for k in shapes if k not in state_dict]
    bad = [k for k in shapes if k in state_dict and tuple(state_dict[k].shape) != tuple(shapes[k])]
    if missing or bad:
        raise ValueError(f"{what}: missing keys {missing[:5]} ({len(missing)}), shape mismatches {bad[:5]} ({len(bad)})")


default_unet_config = lambda: _config.get("sd21_unet")
default_vae_config = lambda: _config.get("sd_vae")
