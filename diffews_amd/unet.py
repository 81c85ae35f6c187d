"""MI355X engine behind the reference's two-input UNet.

Drop-in for `MyUNet2DConditionModel`
(/root/reference/diffews/models/unet_2d_condition.py): same constructor config keys
(U:185-241), `from_pretrained(path, subfolder=)` (evaluation_util/main_oss.py:338-345),
`forward(sample, timestep, encoder_hidden_states, is_target=True, ..., return_dict=True)`
(U:879-895) and `clear_attn_bank()` (U:656-664), with the K/V-bank semantics of `MyAttention`
(diffews/models/attention_processor.py:41-50, 251-267): the first forward after
`clear_attn_bank()` stores every self-attention layer's K/V, the next forward attends over
`[K_own ; K_bank]` with the n-shot batch->token fold of the xformers processor.

All compute is hand-written HIP (diffews_amd/csrc) through the C ABI; activations live in HBM as
NHWC storage-dtype tensors, so the NCHW<->token permutes of Transformer2DModel vanish.
"""
from dataclasses import dataclass

import os

import torch

from . import _lib as L
from . import ops, packing, weights


@dataclass
class UNet2DConditionOutput:
    """Mirror of the reference's output dataclass (U:61-71)."""
    sample: torch.Tensor = None


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _resolve_device(d):
    """torch.device with the index filled in ("cuda" -> "cuda:<current>"), so `.to("cuda:0")` on an engine that
    lives on "cuda" is recognised as a no-op instead of repacking 866 M parameters (E:371 `pipe.to(device)`)."""
    d = torch.device(d)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return d


def _needs_rebuild(self, device, dtype):
    if dtype is not None and dtype != self.dtype:
        return True
    return device is not None and _resolve_device(device) != _resolve_device(self.device)


class _Resnet:
    def __init__(self, sd, p, dev, dt, eps, groups):
        g = lambda k: sd[p + k]
        self.eps, self.groups = eps, groups
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
        self.has_temb = p + "time_emb_proj.weight" in sd
        self.temb_slice = None  # (offset, cout) into the fused time-projection output

    def __call__(self, x, tproj=None):
        """x: the residual stream, NHWC storage dtype -- or fp32 (residual_dtype=torch.float32): GroupNorm then reads
        fp32, conv2 adds the fp32 shortcut and stores fp32; only the branch's MFMA operands are 16-bit."""
        B, H, W, Cin = x.shape
        dt, f32s = self.w1.dtype, x.dtype == torch.float32
        h = ops.groupnorm(x, self.g1, self.b1, self.groups, self.eps, silu=True, out_dtype=dt)
        rb = None
        if self.has_temb:
            o, c = self.temb_slice
            rb = tproj[:, o:o + c]
        h = ops.conv3x3(h, self.w1, self.cout, bias=self.cb1, rowbias=rb, gn_groups=self.groups)
        h = ops.groupnorm(h, self.g2, self.b2, self.groups, self.eps, silu=True)
        sc = x
        if self.ws is not None:
            sc = ops.linear_stream(x.view(-1, Cin), self.ws, bias=self.bs).view(B, H, W, self.cout)
        return ops.conv3x3(h, self.w2, self.cout, bias=self.cb2, residual=sc, gn_groups=self.groups, out_f32=f32s)


class _Transformer:
    """Transformer2DModel(use_linear_projection) with one BasicTransformerBlock; owns the bank."""

    def __init__(self, sd, p, dev, dt, heads, groups):
        g = lambda k: sd[p + k]
        f = lambda k: g(k).float().to(dev)
        w = lambda k: g(k).to(dev, dt).contiguous()
        self.heads, self.groups = heads, groups
        self.gn_g, self.gn_b = f("norm.weight"), f("norm.bias")
        self.w_in, self.b_in = w("proj_in.weight"), f("proj_in.bias")
        self.w_out, self.b_out = w("proj_out.weight"), f("proj_out.bias")
        b = "transformer_blocks.0."
        self.ln = [(f(b + n + ".weight"), f(b + n + ".bias")) for n in ("norm1", "norm2", "norm3")]
        self.w_qkv = torch.cat([g(b + "attn1.to_q.weight"), g(b + "attn1.to_k.weight"),
                                g(b + "attn1.to_v.weight")], 0).to(dev, dt).contiguous()
        self.w_o1, self.b_o1 = w(b + "attn1.to_out.0.weight"), f(b + "attn1.to_out.0.bias")
        self.w_q2 = w(b + "attn2.to_q.weight")
        self.w_kv2 = torch.cat([g(b + "attn2.to_k.weight"), g(b + "attn2.to_v.weight")], 0).to(dev, dt).contiguous()
        self.w_o2, self.b_o2 = w(b + "attn2.to_out.0.weight"), f(b + "attn2.to_out.0.bias")
        wp, bp = packing.pack_geglu(g(b + "ff.net.0.proj.weight"), g(b + "ff.net.0.proj.bias"))
        self.w_ff1, self.b_ff1 = wp.to(dev, dt), bp.to(dev)
        self.w_ff2, self.b_ff2 = w(b + "ff.net.2.weight"), f(b + "ff.net.2.bias")
        self.k_bank = None
        self.v_bank = None
        self.kv_slice = (0, 2 * self.w_q2.shape[0])   # column range in the fused prompt-K/V buffer
        self.fold2 = None   # (G [64, C], U^T [C, 64], L): attn2 folded on a constant prompt (fold_attn2)

    def fold_attn2(self, kv, L_ctx):
        """attn2 on a CONSTANT prompt: with keys/values fixed, per head h and prompt token l
            score[h,l] = LN(x) . g[h,l],   g[h,l] = scale * Wq[h]^T k[l,h]      (one [64, C] matrix G)
            out        = sum_{h,l} softmax_l(score)[h,l] * u[h,l] + b_o,   u[h,l] = Wo[:, h] v[l,h]
        so to_q (C x C), the attention kernel and to_out (C x C) collapse into two thin GEMMs (N = 64 and
        K = 64) around a per-head softmax over the L prompt tokens.  kv: this layer's [L, 2C] slice of the
        folded prompt K/V.  Needs heads * L <= 64 (L = 2 at inference, P:591-600)."""
        C, h = self.w_q2.shape[0], self.heads
        if h * L_ctx > 64:
            self.fold2 = None
            return
        # load-time constant folding on the HOST in fp32 (like the weight repacking): a few hundred
        # [L, 64] x [64, C] products per checkpoint, no device GEMM library involved
        dev = kv.device
        kv = kv.float().cpu()
        k, v = kv[:, :C], kv[:, C:2 * C]                             # [L, C]
        wq, wo = self.w_q2.float().cpu(), self.w_o2.float().cpu()    # [C_out, C_in]
        G = torch.zeros(64, C, dtype=torch.float32)
        Ut = torch.zeros(C, 64, dtype=torch.float32)
        for hh in range(h):
            blk = slice(hh * 64, (hh + 1) * 64)
            G[hh * L_ctx:(hh + 1) * L_ctx] = (64 ** -0.5) * (k[:, blk] @ wq[blk, :])       # [L, C]
            Ut[:, hh * L_ctx:(hh + 1) * L_ctx] = wo[:, blk] @ v[:, blk].t()                # [C, L]
        self.fold2 = (G.to(dev, self.w_q2.dtype).contiguous(), Ut.to(dev, self.w_q2.dtype).contiguous(), L_ctx)

    def clear_bank(self):
        self.k_bank = None
        self.v_bank = None

    def __call__(self, x, ehs2d, L_ctx, n_ref=0):
        """n_ref == 0: reference bank semantics (fill on the first pass after clear, read on the next).
        n_ref > 0: lock-step pair -- the batch is [n_ref support images ; query images]; the support
        rows run plain self-attention and are the bank of the query rows within the same call."""
        B, H, W, C = x.shape
        N = H * W
        heads = self.heads
        # fp32 residual stream (x fp32): the block's running sum t is fp32 too; LayerNorm reads it in fp32, every
        # `+ residual` epilogue adds and stores fp32; q/k/v, attention output, GEGLU and the GEMM operands stay 16-bit
        dt, f32s = self.w_in.dtype, x.dtype == torch.float32
        n = ops.groupnorm(x, self.gn_g, self.gn_b, self.groups, 1e-6, silu=False, out_dtype=dt)
        t = ops.linear(n.view(-1, C), self.w_in, bias=self.b_in, out_f32=f32s)
        # --- attn1: KV-fusion self-attention (A:237-271)
        ln = ops.layernorm(t, *self.ln[0], out_dtype=dt)
        # q leaves the projection multiplied by attn.scale * log2(e) (fp32, before its one rounding): the
        # attention kernel then exponentiates q.k - m directly; k and v (the bank, A:251-267) are untouched
        qkv = ops.linear(ln, self.w_qkv, colscale=(C, ops.FSA_QSCALE)).view(B, N, 3 * C)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        if n_ref:
            bq = B - n_ref
            if bq <= 0 or n_ref % bq != 0:
                raise ValueError(f"{n_ref} support images is not a multiple of the {bq} query images")
            # one launch for both passes: support images attend over their own keys, query images over
            # [own ; their episode's support images]; long (query) workgroups are dispatched first
            att = ops.fsa_attention(q, k, v, heads, k[:n_ref], v[:n_ref], nshot=n_ref // bq, n_plain=n_ref,
                                    q_prescaled=True)
        elif self.k_bank is None:  # A:251-252 / 260-261: first pass after clear fills the bank
            self.k_bank, self.v_bank = k, v
            att = ops.fsa_attention(q, k, v, heads, q_prescaled=True)
        else:                    # A:253-258 / 262-267: [own ; bank], ref batch folded into tokens
            bank_b = self.k_bank.shape[0]
            if bank_b % B != 0:
                raise ValueError(f"bank holds {bank_b} support images, not a multiple of the query batch {B}")
            att = ops.fsa_attention(q, k, v, heads, self.k_bank, self.v_bank, nshot=bank_b // B, q_prescaled=True)
        t = ops.linear(att.view(-1, C), self.w_o1, bias=self.b_o1, residual=t, out_f32=f32s)
        # --- attn2: cross-attention on the prompt tokens
        ln = ops.layernorm(t, *self.ln[1], out_dtype=dt)
        if isinstance(ehs2d, tuple) and len(ehs2d) == 3 and ehs2d[2] and self.fold2 is not None:
            G, Ut, Lf = self.fold2                                   # constant prompt: see fold_attn2
            sc = ops.linear(ln, G, out_f32=True)                     # [M, 64] scores (scale folded in)
            pr = ops.softmax_groups(sc, heads, Lf, dt)               # per-head softmax over the L tokens
            t = ops.linear(pr, Ut, bias=self.b_o2, residual=t, out_f32=f32s)
        else:
            q2 = ops.linear(ln, self.w_q2).view(B, N, C)
            # prompt K/V of all 16 layers come from ONE GEMM per forward (MyUNet2DConditionModel._prompt_kv);
            # this layer's [B, L, 2C] slice is a strided view of that buffer
            o, kv_all = self.kv_slice[0], ehs2d
            if isinstance(kv_all, tuple):
                kv2 = kv_all[1].view(B, L_ctx, -1)[..., o:o + 2 * C]
            else:
                kv2 = ops.linear(ehs2d, self.w_kv2).view(B, L_ctx, 2 * C)
            ca = ops.cross_attention(q2, kv2[..., :C], kv2[..., C:], heads)
            t = ops.linear(ca.view(-1, C), self.w_o2, bias=self.b_o2, residual=t, out_f32=f32s)
        # --- GEGLU feed-forward
        ln = ops.layernorm(t, *self.ln[2], out_dtype=dt)
        ff = ops.linear(ln, self.w_ff1, bias=self.b_ff1, geglu=True)
        # the block's running sum is itself proj_out's operand: fp32 stream => fed as (hi, lo) (ops.linear_stream)
        t = ops.linear(ff, self.w_ff2, bias=self.b_ff2, residual=t, out_f32=f32s)
        return ops.linear_stream(t, self.w_out, bias=self.b_out, residual=x.view(-1, C)).view(B, H, W, C)


class _Conv:
    def __init__(self, sd, p, dev, dt):
        self.w = packing.pack_conv3x3(sd[p + "weight"]).to(dev, dt)
        self.b = sd[p + "bias"].float().to(dev)
        self.cout = sd[p + "weight"].shape[0]


class MyUNet2DConditionModel:
    """HIP engine with the reference UNet's interface: the inference forward (U:879-1258).  The training step of the same
    graph (forward with saved activations + hand-written backward) is `diffews_amd.train.UNetTrainer`.

    residual_dtype: storage of the RESIDUAL STREAM -- the tensor that runs through the blocks as x + branch(x)
    (resnet outputs, transformer residuals, skip tensors, sampler outputs).  None: the storage dtype (fastest).
    torch.float32: the stream is summed and stored in fp32 (GEMM epilogues add an fp32 residual and write fp32,
    GroupNorm / LayerNorm read fp32); MFMA operands stay 16-bit, i.e. each branch sees ONE rounding of its normalised
    input and the stream none.  This is what brings fp16 storage within north_star's 1e-3 of the fp32 reference
    (DESIGN.md section 4)."""

    def __init__(self, config=None, state_dict=None, torch_dtype=torch.bfloat16, device="cuda", residual_dtype=None,
                 **kwargs):
        cfg = weights.default_unet_config()
        cfg.update(config or {})
        cfg.update(kwargs)
        self.config = _Cfg(cfg)
        if residual_dtype not in (None, torch_dtype, torch.float32):
            raise ValueError("residual_dtype must be None (= torch_dtype) or torch.float32")
        self.residual_dtype = residual_dtype or torch_dtype
        self._f32s = self.residual_dtype == torch.float32
        if torch_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError(
                "engine storage dtype must be torch.bfloat16 or torch.float16: the MI355X path keeps activations in 16 bits "
                "(fp32 accumulation, statistics and boundary tensors).  The reference launcher's default is fp32 "
                "(evaluation_util/main_oss.py:335-336): pass --half_precision / torch_dtype=torch.float16 -- z0 then sits "
                "1.5e-3 (relative L2) from the fp32 path, see DESIGN.md section 4")
        self.dtype = torch_dtype
        self.device = torch.device(device)
        L.lib()  # fail loudly right here if the HIP library is not built
        if state_dict is None:
            raise ValueError("MyUNet2DConditionModel needs a state_dict (use from_pretrained / synthetic weights)")
        weights.check_state_dict(state_dict, weights.unet_param_shapes(cfg), "unet")
        self._build(state_dict)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, path, subfolder=None, revision=None, torch_dtype=torch.bfloat16, device="cuda",
                        residual_dtype=None, **kw):
        cfg = weights.load_config(path, subfolder)
        sd = weights.load_state_dict(path, subfolder)
        return cls(cfg, sd, torch_dtype=torch_dtype, device=device, residual_dtype=residual_dtype)

    def save_pretrained(self, path, subfolder=None):
        weights.save_pretrained(path, dict(self.config), self._sd_cpu, subfolder)

    def _build(self, sd):
        cfg, dev, dt = self.config, self.device, self.dtype
        self._sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
        boc = list(cfg["block_out_channels"])
        heads = cfg["attention_head_dim"]
        heads = [heads] * len(boc) if isinstance(heads, int) else list(heads)
        for c, h in zip(boc, heads):
            if c % 64 or c // h != 64:
                raise ValueError("gfx950 kernels need channel counts that are multiples of 64 and head_dim 64")
        groups, eps, lpb = cfg["norm_num_groups"], cfg["norm_eps"], cfg["layers_per_block"]
        self.groups, self.eps = groups, eps
        # boundary convs (Cin <= 8): fp32 weights for the direct kernel
        self.w_in = packing.pack_conv_small(sd["conv_in.weight"]).to(dev)
        self.b_in = sd["conv_in.bias"].float().to(dev)
        self.w_in_ref = packing.pack_conv_small(sd["conv_in_ref.weight"]).to(dev)
        self.b_in_ref = sd["conv_in_ref.bias"].float().to(dev)
        self.te_w1 = sd["time_embedding.linear_1.weight"].to(dev, dt).contiguous()
        self.te_b1 = sd["time_embedding.linear_1.bias"].float().to(dev)
        self.te_w2 = sd["time_embedding.linear_2.weight"].to(dev, dt).contiguous()
        self.te_b2 = sd["time_embedding.linear_2.bias"].float().to(dev)
        R = lambda p: _Resnet(sd, p, dev, dt, eps, groups)
        T = lambda p, h: _Transformer(sd, p, dev, dt, h, groups)
        self.down = []
        for i, typ in enumerate(cfg["down_block_types"]):
            blk = dict(res=[R(f"down_blocks.{i}.resnets.{j}.") for j in range(lpb)], attn=None, down=None)
            if typ == "CrossAttnDownBlock2D":
                blk["attn"] = [T(f"down_blocks.{i}.attentions.{j}.", heads[i]) for j in range(lpb)]
            if i != len(boc) - 1:
                blk["down"] = _Conv(sd, f"down_blocks.{i}.downsamplers.0.conv.", dev, dt)
            self.down.append(blk)
        self.mid = dict(res=[R("mid_block.resnets.0."), R("mid_block.resnets.1.")],
                        attn=T("mid_block.attentions.0.", heads[-1]))
        rheads = heads[::-1]
        self.up = []
        for i, typ in enumerate(cfg["up_block_types"]):
            blk = dict(res=[R(f"up_blocks.{i}.resnets.{j}.") for j in range(lpb + 1)], attn=None, up=None)
            if typ == "CrossAttnUpBlock2D":
                blk["attn"] = [T(f"up_blocks.{i}.attentions.{j}.", rheads[i]) for j in range(lpb + 1)]
            if i != len(boc) - 1:
                blk["up"] = _Conv(sd, f"up_blocks.{i}.upsamplers.0.conv.", dev, dt)
            self.up.append(blk)
        self.gn_out = (sd["conv_norm_out.weight"].float().to(dev), sd["conv_norm_out.bias"].float().to(dev))
        self.conv_out = _Conv(sd, "conv_out.", dev, dt)
        # all 22 time_emb_proj layers fused into one GEMM: [sum(Cout), temb]
        ws, bs, off = [], [], 0
        for r, p in self._resnets_with_prefix():
            ws.append(sd[p + "time_emb_proj.weight"])
            bs.append(sd[p + "time_emb_proj.bias"])
            r.temb_slice = (off, r.cout)
            off += r.cout
        self.tp_w = torch.cat(ws, 0).to(dev, dt).contiguous()
        self.tp_b = torch.cat(bs, 0).float().to(dev)
        # prompt K/V projections (attn2.to_k / to_v) of every transformer layer fused into one GEMM
        kvw, off = [], 0
        for t in self._transformers():
            kvw.append(t.w_kv2)
            t.kv_slice = (off, t.w_kv2.shape[0])
            off += t.w_kv2.shape[0]
        self.kv_w_all = torch.cat(kvw, 0).contiguous()
        # fold_conditioning() also folds attn2 on the constant prompt (set False before folding for an A/B run)
        self.fold_attn2 = True

    def _resnets_with_prefix(self):
        for i, blk in enumerate(self.down):
            for j, r in enumerate(blk["res"]):
                yield r, f"down_blocks.{i}.resnets.{j}."
        yield self.mid["res"][0], "mid_block.resnets.0."
        yield self.mid["res"][1], "mid_block.resnets.1."
        for i, blk in enumerate(self.up):
            for j, r in enumerate(blk["res"]):
                yield r, f"up_blocks.{i}.resnets.{j}."

    def _transformers(self):
        for blk in self.down + self.up:
            for t in blk["attn"] or []:
                yield t
        yield self.mid["attn"]

    # ------------------------------------------------------------------ reference API
    def clear_attn_bank(self):
        for t in self._transformers():
            t.clear_bank()

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):      # nn.Module.to(dtype) call form
            device, dtype = None, device
        if _needs_rebuild(self, device, dtype):
            self.__init__(dict(self.config), self._sd_cpu, torch_dtype=dtype or self.dtype,
                          device=device or self.device,
                          residual_dtype=torch.float32 if self._f32s else None)
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None  # the HIP attention kernel is always the memory-efficient one

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, is_target=True, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict=True,
                out_scale=1.0):
        for name, v in (("class_labels", class_labels), ("timestep_cond", timestep_cond),
                        ("attention_mask", attention_mask), ("added_cond_kwargs", added_cond_kwargs),
                        ("down_block_additional_residuals", down_block_additional_residuals),
                        ("mid_block_additional_residual", mid_block_additional_residual),
                        ("down_intrablock_additional_residuals", down_intrablock_additional_residuals),
                        ("encoder_attention_mask", encoder_attention_mask)):
            if v is not None:
                raise NotImplementedError(f"{name} is not on the DiffewS hot path (always None there)")
        cfg, dt, dev = self.config, self.dtype, self.device
        in_dtype = sample.dtype
        x_in = sample.to(device=dev, dtype=torch.float32).contiguous()
        B, Cin, h, w = x_in.shape
        # ---- 1. time (U:991-1015)
        c0 = cfg["block_out_channels"][0]
        kv_all = None
        if encoder_hidden_states is None:     # folded conditioning (fold_conditioning)
            tproj, ehs2d, kv_all, L_ctx = self._folded_rows(B, timestep)
        else:
            tproj = self._time_proj(B, timestep)  # [B, sum Cout] fp32: all 22 time_emb_proj outputs
            # ---- prompt tokens
            ehs = encoder_hidden_states.to(device=dev, dtype=dt)
            if ehs.shape[0] != B:
                raise ValueError("encoder_hidden_states batch must match sample batch")
            L_ctx = ehs.shape[1]
            ehs2d = ehs.reshape(B * L_ctx, ehs.shape[2]).contiguous()
        # ---- 2. conv_in | conv_in_ref (U:1117-1121)
        if is_target:
            if Cin != cfg["in_channels"]:
                raise ValueError(f"target pass expects {cfg['in_channels']} channels, got {Cin}")
            x = ops.conv_small(x_in, self.w_in, self.b_in, c0, 9, dt, out_f32=self._f32s)
        else:
            if Cin != cfg["in_channels_ref"]:
                raise ValueError(f"support pass expects {cfg['in_channels_ref']} channels, got {Cin}")
            x = ops.conv_small(x_in, self.w_in_ref, self.b_in_ref, c0, 9, dt, out_f32=self._f32s)
        out = self._trunk(x, tproj, ehs2d, L_ctx, 0, out_scale, kv_all)
        if in_dtype in (torch.float16, torch.bfloat16, torch.float64):
            out = out.to(in_dtype)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    def _time_proj(self, B, timestep):
        cfg, dt, dev = self.config, self.dtype, self.device
        if torch.is_tensor(timestep) and timestep.device.type == "cpu" and timestep.numel() == 1:
            timestep = float(timestep)
        if not torch.is_tensor(timestep):
            t = torch.full((B,), float(timestep), dtype=torch.float32, device=dev)
        else:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        temb = ops.timestep_embedding(t, cfg["block_out_channels"][0], dt, cfg["flip_sin_to_cos"], float(cfg["freq_shift"]))
        e = ops.linear(temb, self.te_w1, bias=self.te_b1, act=L.ACT_SILU)
        semb = ops.linear(e, self.te_w2, bias=self.te_b2, act=L.ACT_SILU)
        return ops.linear(semb, self.tp_w, bias=self.tp_b, out_f32=True)

    # ------------------------------------------------------------------ constant folding
    @torch.no_grad()
    def fold_conditioning(self, timestep, prompt_embed):
        """The prompt ("" through CLIP, P:585-601) and the single timestep (S:107-180 -> [1]) are
        constants of a checkpoint, hence so are the timestep embedding (U:991-1015), all 22 resnet
        time_emb_proj outputs and all 16 attn2 K/V projections.  Compute them ONCE here; forwards
        called with encoder_hidden_states=None (and the same timestep) then start at conv_in with no
        conditioning kernels at all.  prompt_embed: [1, L, cross_attention_dim] or [L, D]."""
        dt, dev = self.dtype, self.device
        pe = prompt_embed.to(device=dev, dtype=dt)
        pe = pe.reshape(-1, pe.shape[-1]).contiguous()
        t = float(timestep)
        self._folded = {"t": t, "L": pe.shape[0], "prompt": pe, "tproj": self._time_proj(1, t),
                        "kv": ops.linear(pe, self.kv_w_all), "rows": {}}
        for tr in self._transformers():
            tr.fold2 = None
            if self.fold_attn2:
                o, n = tr.kv_slice
                tr.fold_attn2(self._folded["kv"][:, o:o + n], pe.shape[0])
        return self

    def unfold_conditioning(self):
        self._folded = None
        for tr in self._transformers():
            tr.fold2 = None

    def _folded_rows(self, B, timestep):
        f = getattr(self, "_folded", None)
        if f is None:
            raise ValueError("encoder_hidden_states=None needs fold_conditioning() first")
        if torch.is_tensor(timestep):
            if timestep.device.type != "cpu" or timestep.numel() != 1:
                raise ValueError("folded conditioning takes the timestep as a python number or a CPU scalar")
        if float(timestep) != f["t"]:
            raise ValueError(f"conditioning was folded for timestep {f['t']}, got {float(timestep)}")
        if B not in f["rows"]:   # materialised once per batch size (rowbias / K/V rows are per image)
            f["rows"][B] = (f["tproj"].expand(B, -1).contiguous(), f["prompt"].repeat(B, 1).contiguous(),
                            f["kv"].repeat(B, 1).contiguous())
        tproj, ehs2d, kv = f["rows"][B]
        return tproj, ehs2d, kv, f["L"]

    @torch.no_grad()
    def forward_pair(self, ref_sample, query_sample, timestep, ehs_ref=None, ehs_query=None, out_scale=1.0):
        """Support and query passes in layer lock-step: ONE trunk pass over the batch
        [support images ; query images] (weights read once, twice the rows per GEMM).  Per image the
        arithmetic is that of forward(ref, is_target=False) followed by forward(query): every op on
        the path is per-image (GroupNorm) or per-token, and each query image attends over
        [own ; its episode's support images] exactly as with the bank (A:251-267).
        Returns the query pass' output only (the reference discards the support pass' output, P:719)."""
        cfg, dt, dev = self.config, self.dtype, self.device
        zr = ref_sample.to(device=dev, dtype=torch.float32).contiguous()
        zq = query_sample.to(device=dev, dtype=torch.float32).contiguous()
        n_ref, bq = zr.shape[0], zq.shape[0]
        if zr.shape[1] != cfg["in_channels_ref"] or zq.shape[1] != cfg["in_channels"]:
            raise ValueError("forward_pair expects (in_channels_ref, in_channels) channel counts")
        c0 = cfg["block_out_channels"][0]
        kv_all = None
        if ehs_ref is None and ehs_query is None:
            tproj, ehs2d, kv_all, L_ctx = self._folded_rows(n_ref + bq, timestep)
        else:
            tproj = self._time_proj(n_ref + bq, timestep)
            ehs = torch.cat([ehs_ref.to(device=dev, dtype=dt), ehs_query.to(device=dev, dtype=dt)], dim=0)
            L_ctx = ehs.shape[1]
            ehs2d = ehs.reshape((n_ref + bq) * L_ctx, ehs.shape[2]).contiguous()
        x = torch.empty(n_ref + bq, zq.shape[2], zq.shape[3], c0, dtype=self.residual_dtype, device=dev)
        ops.conv_small(zr, self.w_in_ref, self.b_in_ref, c0, 9, dt, out=x[:n_ref], out_f32=self._f32s)   # conv_in_ref (U:1119)
        ops.conv_small(zq, self.w_in, self.b_in, c0, 9, dt, out=x[n_ref:], out_f32=self._f32s)           # conv_in     (U:1121)
        out = self._trunk(x, tproj, ehs2d, L_ctx, n_ref, out_scale, kv_all)
        return out[n_ref:]

    def _trunk(self, x, tproj, ehs2d, L_ctx, n_ref, out_scale, kv_all=None):
        # all layers' prompt K/V in one launch: [B*L, sum(2C)]; layers take column slices; the third
        # entry says the prompt is the folded constant (layers may then use their folded attn2)
        ehs2d = (ehs2d, kv_all if kv_all is not None else ops.linear(ehs2d, self.kv_w_all), kv_all is not None)
        # ---- 3. down (U:1153-1175)
        skips = [x]
        for blk in self.down:
            for j, r in enumerate(blk["res"]):
                x = r(x, tproj)
                if blk["attn"] is not None:
                    x = blk["attn"][j](x, ehs2d, L_ctx, n_ref)
                skips.append(x)
            if blk["down"] is not None:
                d = blk["down"]
                x = ops.conv3x3_stream(x, d.w, d.cout, bias=d.b, stride=2, pad=1)
                skips.append(x)
        # ---- 4. mid (U:1189-1198)
        x = self.mid["res"][0](x, tproj)
        x = self.mid["attn"](x, ehs2d, L_ctx, n_ref)
        x = self.mid["res"][1](x, tproj)
        # ---- 5. up (U:1214-1243)
        for blk in self.up:
            for j, r in enumerate(blk["res"]):
                x = ops.concat_channels(x, skips.pop())
                x = r(x, tproj)
                if blk["attn"] is not None:
                    x = blk["attn"][j](x, ehs2d, L_ctx, n_ref)
            if blk["up"] is not None:
                u = blk["up"]
                x = ops.conv3x3_stream(x, u.w, u.cout, bias=u.b, ups=True)
        # ---- 6. out (U:1246-1249); out_scale lets the pipeline fold z0 = -v into the epilogue
        x = ops.groupnorm(x, *self.gn_out, self.groups, self.eps, silu=True, out_dtype=self.dtype)
        co = self.conv_out
        return ops.conv3x3(x, co.w, co.cout, bias=co.b, out_nchw_f32=True, out_scale=out_scale)


CustomUNet2DConditionModel = MyUNet2DConditionModel  # name used by evaluation_util/main_oss.py:27
