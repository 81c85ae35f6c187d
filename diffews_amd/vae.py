"""MI355X engine behind diffusers' AutoencoderKL as the reference pipeline drives it:
`vae.encoder(x)` + `vae.quant_conv(h)` (marigold_pipeline_rgb_latent_noise.py:852-853) and
`vae.post_quant_conv(z)` + `vae.decoder(z)` (P:901-902); also `encode(x).latent_dist` for the
training launcher (train_icl_multitask_nocrop_nearest_nshot_v3.py:1347).

Boundary tensors are NCHW fp32 like the reference's; inside, activations are NHWC storage dtype.
"""
import torch

from . import _lib as L
from . import ops, packing, weights
from .unet import _Cfg, _Conv, _needs_rebuild


class _VaeResnet:
    def __init__(self, sd, p, dev, dt, groups):
        g = lambda k: sd[p + k]
        self.groups = groups
        self.g1, self.b1 = g("norm1.weight").float().to(dev), g("norm1.bias").float().to(dev)
        self.g2, self.b2 = g("norm2.weight").float().to(dev), g("norm2.bias").float().to(dev)
        self.cout = g("conv1.weight").shape[0]
        self.w1 = packing.pack_conv3x3(g("conv1.weight")).to(dev, dt)
        self.cb1 = g("conv1.bias").float().to(dev)
        self.w2 = packing.pack_conv3x3(g("conv2.weight")).to(dev, dt)
        self.cb2 = g("conv2.bias").float().to(dev)
        self.ws = None
        if p + "conv_shortcut.weight" in sd:
            self.ws = packing.pack_conv1x1(g("conv_shortcut.weight")).to(dev, dt)
            self.bs = g("conv_shortcut.bias").float().to(dev)

    def __call__(self, x):
        """x: the residual stream (storage dtype, or fp32 with residual_dtype=torch.float32: see unet._Resnet)."""
        B, H, W, Cin = x.shape
        dt, f32s = self.w1.dtype, x.dtype == torch.float32
        # gn_in: GroupNorm + SiLU pass, then the conv; gn_groups: the conv's epilogue leaves the statistics for the next norm
        h = ops.conv3x3(x, self.w1, self.cout, bias=self.cb1, gn_groups=self.groups,
                        gn_in=(self.g1, self.b1, self.groups, 1e-6, True))
        sc = x
        if self.ws is not None:
            sc = ops.linear_stream(x.view(-1, Cin), self.ws, bias=self.bs).view(B, H, W, self.cout)
        return ops.conv3x3(h, self.w2, self.cout, bias=self.cb2, residual=sc, gn_groups=self.groups,
                           gn_in=(self.g2, self.b2, self.groups, 1e-6, True), out_f32=f32s)


class _VaeAttention:
    """Mid-block attention: GN -> biased q/k/v -> 1 head of dim C -> out proj -> + residual."""

    def __init__(self, sd, p, dev, dt, groups):
        f = lambda k: sd[p + k].float().to(dev)
        w = lambda k: sd[p + k].to(dev, dt).contiguous()
        self.groups = groups
        self.gn = (f("group_norm.weight"), f("group_norm.bias"))
        self.wq, self.bq = w("to_q.weight"), f("to_q.bias")
        self.wk, self.bk = w("to_k.weight"), f("to_k.bias")
        self.wv, self.bv = w("to_v.weight"), f("to_v.bias")
        self.wo, self.bo = w("to_out.0.weight"), f("to_out.0.bias")
        # fused [Wq; Wk; Wv] projection (one GEMM, N = 3 C) feeding the flash kernel (round 4)
        self.wqkv = torch.cat([self.wq, self.wk, self.wv], 0).contiguous()
        self.bqkv = torch.cat([self.bq, self.bk, self.bv], 0).contiguous()
        # "auto": flash when the fp32 scores of the batch would exceed 1 GiB, or after enable_xformers_memory_efficient_attention()
        # (E:374-376) -- measured on MI355X (12 x 4096 tokens): flash 0.90 ms / no N x N tensor, materialised 0.85 ms / 1.2 GB of
        # scores + probabilities; at 16 images flash wins (0.92 vs 1.04 ms).  True / False force one path.
        self.flash = "auto"

    def __call__(self, x):
        B, H, W, C = x.shape
        N = H * W
        dt, f32s = self.wq.dtype, x.dtype == torch.float32
        n = ops.groupnorm(x, *self.gn, self.groups, 1e-6, silu=False, out_dtype=dt).view(-1, C)
        # "auto": the materialised path while its fp32 scores fit 1 GiB and the token count fits its 64-wide K chunks (it is
        # 0.4 ms faster on the headline episode); the flash kernel beyond that, ragged token counts included
        use_flash = self.flash is True or (self.flash == "auto" and (B * N * N * 4 > (1 << 30) or N % 64 != 0))
        if use_flash and C == 512:
            # the reference enables xformers' memory-efficient attention for the VAE too (E:374-376): the N x N scores are
            # never materialised.  One fused QKV GEMM (q leaves it pre-scaled by C^-0.5 * log2 e, in fp32 before its single
            # rounding), one flash kernel (csrc/vae_attention.hip), the output projection with the residual.
            qkv = ops.linear(n, self.wqkv, bias=self.bqkv, colscale=(C, ops.VATTN_QSCALE)).view(B, N, 3 * C)
            o = ops.vae_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:])
            return ops.linear(o.view(-1, C), self.wo, bias=self.bo, residual=x.view(-1, C), out_f32=f32s).view(B, H, W, C)
        q = ops.linear(n, self.wq, bias=self.bq).view(B, N, C)
        k = ops.linear(n, self.wk, bias=self.bk).view(B, N, C)
        v = ops.linear(n, self.wv, bias=self.bv).view(B, N, C)
        s = ops.bmm_nt(q, k, out_f32=True)                       # [B, N, N] fp32 scores
        p = ops.softmax_rows(s, dt, scale=C ** -0.5)             # upcast softmax, storage-dtype probs
        o = ops.bmm_nt(p, ops.transpose(v))                      # [B, N, C]
        return ops.linear(o.view(-1, C), self.wo, bias=self.bo, residual=x.view(-1, C), out_f32=f32s).view(B, H, W, C)


class _Mid:
    def __init__(self, sd, p, dev, dt, groups):
        self.r0 = _VaeResnet(sd, p + "resnets.0.", dev, dt, groups)
        self.att = _VaeAttention(sd, p + "attentions.0.", dev, dt, groups)
        self.r1 = _VaeResnet(sd, p + "resnets.1.", dev, dt, groups)

    def __call__(self, x):
        return self.r1(self.att(self.r0(x)))


class _Encoder:
    def __init__(self, vae, sd):
        cfg, dev, dt, g = vae.config, vae.device, vae.dtype, vae.config["norm_num_groups"]
        self.dt, self.groups, self.f32s = dt, g, vae.residual_dtype == torch.float32
        self.single_round_levels = 2      # downsamplers of the first N levels read the fp32 stream as ONE rounded operand
        boc, lpb = list(cfg["block_out_channels"]), cfg["layers_per_block"]
        self.c0 = boc[0]
        self.w_in = packing.pack_conv_small(sd["encoder.conv_in.weight"]).to(dev)
        self.b_in = sd["encoder.conv_in.bias"].float().to(dev)
        self.blocks = []
        for i in range(len(boc)):
            res = [_VaeResnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", dev, dt, g) for j in range(lpb)]
            down = _Conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv.", dev, dt) if i != len(boc) - 1 else None
            self.blocks.append((res, down))
        self.mid = _Mid(sd, "encoder.mid_block.", dev, dt, g)
        self.gn_out = (sd["encoder.conv_norm_out.weight"].float().to(dev),
                       sd["encoder.conv_norm_out.bias"].float().to(dev))
        self.conv_out = _Conv(sd, "encoder.conv_out.", dev, dt)

    @torch.no_grad()
    def __call__(self, x):
        """x: [n, 3, H, W], or a list of up to three such tensors encoded as ONE batch (their concatenation,
        read in place by conv_in -- the support images, support masks and query images of P:649-651)."""
        if isinstance(x, (list, tuple)):
            x = [t.to(dtype=torch.float32).contiguous() for t in x]
        else:
            x = x.to(dtype=torch.float32).contiguous()
        h = ops.conv_small(x, self.w_in, self.b_in, self.c0, 9, self.dt, gn_groups=self.groups, out_f32=self.f32s)
        for lvl, (res, down) in enumerate(self.blocks):
            for r in res:
                h = r(h)
            if down is not None:  # F.pad(0,1,0,1) + conv stride 2 padding 0
                # fp32 stream: the two high-resolution downsamplers read it rounded once (ops.conv3x3_stream lo=False), the rest
                # as (hi, lo) pairs: see DESIGN.md section 4 for the error / time table that put the line there
                h = ops.conv3x3_stream(h, down.w, down.cout, bias=down.b, stride=2, pad=0, gn_groups=self.groups,
                                       lo=lvl >= self.single_round_levels)
        h = self.mid(h)
        h = ops.groupnorm(h, *self.gn_out, self.groups, 1e-6, silu=True, out_dtype=self.dt)
        co = self.conv_out
        return ops.conv3x3(h, co.w, co.cout, bias=co.b, out_nchw_f32=True)  # [n, 2*lc, h, w] fp32


class _Decoder:
    def __init__(self, vae, sd):
        cfg, dev, dt, g = vae.config, vae.device, vae.dtype, vae.config["norm_num_groups"]
        self.dt, self.groups, self.f32s = dt, g, (vae.residual_dtype == torch.float32 and vae.decoder_f32_stream)
        boc, lpb = list(cfg["block_out_channels"]), cfg["layers_per_block"]
        rboc = boc[::-1]
        self.c0 = rboc[0]
        self.w_in = packing.pack_conv_small(sd["decoder.conv_in.weight"]).to(dev)
        self.b_in = sd["decoder.conv_in.bias"].float().to(dev)
        self.mid = _Mid(sd, "decoder.mid_block.", dev, dt, g)
        self.blocks = []
        for i in range(len(boc)):
            res = [_VaeResnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", dev, dt, g) for j in range(lpb + 1)]
            up = _Conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv.", dev, dt) if i != len(boc) - 1 else None
            self.blocks.append((res, up))
        self.gn_out = (sd["decoder.conv_norm_out.weight"].float().to(dev),
                       sd["decoder.conv_norm_out.bias"].float().to(dev))
        self.conv_out = _Conv(sd, "decoder.conv_out.", dev, dt)

    @torch.no_grad()
    def __call__(self, z, clamp=False):
        """clamp: clip the image to [-1, 1] in the last conv's epilogue (decode_seg, P:903)."""
        z = z.to(dtype=torch.float32).contiguous()
        h = ops.conv_small(z, self.w_in, self.b_in, self.c0, 9, self.dt, gn_groups=self.groups, out_f32=self.f32s)
        h = self.mid(h)
        for res, up in self.blocks:
            for r in res:
                h = r(h)
            if up is not None:
                h = ops.conv3x3_stream(h, up.w, up.cout, bias=up.b, ups=True, gn_groups=self.groups)
        h = ops.groupnorm(h, *self.gn_out, self.groups, 1e-6, silu=True, out_dtype=self.dt)
        co = self.conv_out
        return ops.conv3x3(h, co.w, co.cout, bias=co.b, out_nchw_f32=True,
                           act=L.ACT_CLAMP1 if clamp else L.ACT_NONE)  # [b, 3, H, W] fp32


class _Conv1x1Boundary:
    """quant_conv / post_quant_conv: 1x1 conv on NCHW fp32 latents."""

    def __init__(self, sd, p, dev, dt):
        self.w = packing.pack_conv_small(sd[p + "weight"]).to(dev)
        self.b = sd[p + "bias"].float().to(dev)
        self.cout, self.dt = sd[p + "weight"].shape[0], dt
        self._head = {}

    @torch.no_grad()
    def __call__(self, x, in_scale=1.0, out_scale=1.0, out=None, channels=None):
        """channels: compute only the first `channels` outputs (the latent MEAN of quant_conv's moments,
        P:858-861); out: NCHW fp32 view to write into (may be a channel slice of a wider tensor)."""
        w, b, cout = self.w, self.b, self.cout
        if channels is not None and channels != cout:
            if channels not in self._head:
                self._head[channels] = (w[:channels].contiguous(), b[:channels].contiguous())
            (w, b), cout = self._head[channels], channels
        return ops.conv_small(x.to(dtype=torch.float32).contiguous(), w, b, cout, 1, self.dt,
                              nchw_f32_out=True, in_scale=in_scale, out_scale=out_scale, out=out)


class _DiagonalGaussian:
    def __init__(self, moments):
        self.mean, self.logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = self.logvar.clamp(-30.0, 20.0)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + torch.exp(0.5 * self.logvar) * noise


class _EncOut:
    def __init__(self, dist):
        self.latent_dist = dist


class AutoencoderKL:
    """residual_dtype: None (= torch_dtype) or torch.float32 -- storage of the residual stream, see
    MyUNet2DConditionModel.  The encoder carries 1.3 of the 1.45e-3 fp16 error of the 16-bit stream.
    decoder_f32_stream (round 4, default False): the fp32 stream applies to the ENCODER only.  The decoder is downstream of
    the parity tensor (the predicted latent z0, P:769) and its output is quantised to uint8 (P:534), where the 16-bit
    stream's error is 0.15 levels; running it with the fp32 stream bought nothing measurable and cost ~4 ms per step."""

    def __init__(self, config=None, state_dict=None, torch_dtype=torch.bfloat16, device="cuda", residual_dtype=None,
                 decoder_f32_stream=False, **kwargs):
        cfg = weights.default_vae_config()
        cfg.update(config or {})
        cfg.update(kwargs)
        self.config = _Cfg(cfg)
        if residual_dtype not in (None, torch_dtype, torch.float32):
            raise ValueError("residual_dtype must be None (= torch_dtype) or torch.float32")
        self.residual_dtype = residual_dtype or torch_dtype
        self.decoder_f32_stream = bool(decoder_f32_stream)
        if torch_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError(
                "engine storage dtype must be torch.bfloat16 or torch.float16: the MI355X path keeps activations in 16 bits "
                "(fp32 accumulation, statistics and boundary tensors).  The reference launcher's default is fp32 "
                "(evaluation_util/main_oss.py:335-336): pass --half_precision / torch_dtype=torch.float16 -- z0 then sits "
                "1.5e-3 (relative L2) from the fp32 path, see DESIGN.md section 4")
        self.dtype, self.device = torch_dtype, torch.device(device)
        L.lib()
        if state_dict is None:
            raise ValueError("AutoencoderKL needs a state_dict (use from_pretrained / synthetic weights)")
        weights.check_state_dict(state_dict, weights.vae_param_shapes(cfg), "vae")
        for c in cfg["block_out_channels"]:
            if c % 64:
                raise ValueError("gfx950 kernels need VAE channel counts that are multiples of 64")
        self._sd_cpu = {k: v.detach().cpu() for k, v in state_dict.items()}
        sd = state_dict
        self.encoder = _Encoder(self, sd)
        self.decoder = _Decoder(self, sd)
        self.quant_conv = _Conv1x1Boundary(sd, "quant_conv.", self.device, self.dtype)
        self.post_quant_conv = _Conv1x1Boundary(sd, "post_quant_conv.", self.device, self.dtype)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=torch.bfloat16, device="cuda", residual_dtype=None, **kw):
        return cls(weights.load_config(path, subfolder), weights.load_state_dict(path, subfolder),
                   torch_dtype=torch_dtype, device=device, residual_dtype=residual_dtype)

    def save_pretrained(self, path, subfolder=None):
        weights.save_pretrained(path, dict(self.config), self._sd_cpu, subfolder)

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if _needs_rebuild(self, device, dtype):
            self.__init__(dict(self.config), self._sd_cpu, torch_dtype=dtype or self.dtype,
                          device=device or self.device,
                          residual_dtype=torch.float32 if self.residual_dtype == torch.float32 else None,
                          decoder_f32_stream=self.decoder_f32_stream)
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def encode(self, x):
        return _EncOut(_DiagonalGaussian(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        class _Out:
            pass
        o = _Out()
        o.sample = self.decoder(self.post_quant_conv(z))
        return o
