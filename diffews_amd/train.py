"""Training step of the two-input UNet on the MI355X engine (BASELINE configs[4]).

Counterpart of the hot loop of
/root/reference/train_tools/train_icl_multitask_nocrop_nearest_nshot_v3.py (= T):
    T:1374  unet(z_refcat, t, ehs_nshot, is_target=False)    support pass, WITH grad: the K/V banks keep the graph
    T:1375  unet(z_tag, t, ehs, is_target=True)              query pass over [own ; bank] (A:251-267)
    T:1384  loss = F.mse_loss(pred.float(), target.float())  target = -z_mask_tag (T:1371)
    T:1391  backward   T:1393 clip_grad_norm_(1.0)   T:1394 AdamW step (T:1186-1194)
as ONE lock-step trunk pass over the batch [support images ; query images] (same arithmetic per image as the two
passes, see unet.forward_pair), a hand-written backward through the same graph, and a fused AdamW on flat fp32
buffers.  All compute is libdiffews_hip.so (ops.py / ops_bwd.py); torch is device memory, streams and -- once per
optimizer step -- nothing else: the 16-bit shadow is written by the AdamW kernel and the transposed / tap-mirrored copies the
data-gradient GEMMs read by dfw_weight_relayout.

Parameters live in ONE flat fp32 master buffer in the engine's PACKED layouts (conv [Cout][ky][kx][Cin], fused
[Wq;Wk;Wv], fused prompt [Wk2;Wv2] of all layers, the 22 time_emb_proj layers as one matrix, GEGLU rows interleaved),
so weight gradients are produced directly in place by the TN GEMM and the optimizer is layout-agnostic;
`state_dict()` / `grad_dict()` convert to the diffusers key layout (checkpoint format, T:1130-1140; parity tests).
"""
import math

import torch

from . import _lib as L
from . import ops, packing, weights
from . import ops_bwd as ob


class _Tape:
    """Reverse-mode tape over engine tensors: ops register (output, closure).  Gradients are keyed by the tensor's
    MEMORY (data pointer + element count), so a reshaped view handed to the next op routes its gradient to the op that
    produced the storage; a tensor consumed more than once (residuals, skips, banks) has its gradients summed by the
    library's add kernel; a closure receives the gradient in the shape of the output it was registered with."""

    def __init__(self):
        self.ops, self.g = [], {}

    @staticmethod
    def _key(t):
        return (t.data_ptr(), t.numel())

    def add(self, out, fn):
        self.ops.append((out, fn))

    def accum(self, t, g):
        k = self._key(t)
        have = self.g.get(k)
        self.g[k] = g if have is None else ob.add(have, g.reshape(have.shape))

    def take(self, t):
        """Remove and return the gradient t has received so far (None if none, or if it cannot be handed to a kernel as a
        plain operand): a norm backward adds it inside its kernel (dx_add) instead of a separate add pass."""
        have = self.g.get(self._key(t))
        if have is None or not have.is_contiguous() or have.dtype != t.dtype:
            return None
        return self.g.pop(self._key(t))

    def backward(self, out, dout, after=None):
        """after: called once after every closure that ran (the overlapped gradient all-reduce looks there for parameter
        gradients that just became final)."""
        self.accum(out, dout)
        for o, fn in reversed(self.ops):
            g = self.g.pop(self._key(o), None)
            if g is not None:
                fn(g.reshape(o.shape))
                if after is not None:
                    after()
        self.ops, self.g = [], {}


class ParamStore:
    """Flat fp32 master / gradient buffers + a 16-bit shadow, addressed by packed-parameter name."""

    def __init__(self, device, dtype):
        self.device, self.dtype = torch.device(device), dtype
        self.spec, self._init, self.numel = {}, [], 0

    def add(self, name, tensor):
        t = tensor.detach().float().contiguous()
        n = (t.numel() + 63) // 64 * 64              # 256-byte aligned entries
        self.spec[name] = (self.numel, tuple(t.shape))
        self._init.append((self.numel, t))
        self.numel += n

    TAIL = 64   # floats behind the last parameter gradient: slot 0 carries the step's loss through the last gradient bucket

    def finalize(self):
        self.master = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        for off, t in self._init:
            self.master[off:off + t.numel()].copy_(t.reshape(-1))
        self._init = None
        # gradient buffer + TAIL: `grad` (what the optimizer, the norm and the exports see) is the first numel floats; the
        # collective's last bucket extends over the tail, so the scalar loss is averaged over the ranks with it (T:1387)
        self.grad_buf = torch.zeros(self.numel + self.TAIL, dtype=torch.float32, device=self.device)
        self.grad = self.grad_buf[:self.numel]
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.shadow = torch.empty(self.numel, dtype=self.dtype, device=self.device)
        self.version = -1
        self.sync_shadow()
        self._written, self._new = set(), []

    def _view(self, buf, name):
        off, shape = self.spec[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    def p(self, name):      # fp32 master view (biases, norm parameters, boundary-conv weights are read as fp32)
        return self._view(self.master, name)

    def w(self, name):      # storage-dtype shadow view (MFMA operands)
        return self._view(self.shadow, name)

    def g(self, name):
        return self._view(self.grad, name)

    # First-touch gradient writes: a step that starts from zero gradients does not memset the 3.5 GB flat buffer; the first
    # kernel that writes a parameter's gradient stores (accumulate = 0), later ones add.  The alignment pads between
    # entries are never written and keep their zeros; entries nobody touched are zeroed at the end of the step.
    def begin_step(self, fresh):
        self._touched = set() if fresh else None
        self._written, self._new = set(), []

    def acc(self, *names):
        """accumulate flag for a kernel about to write these gradients (all or none must have been written before).
        Every parameter's gradient is written by exactly ONE launch per micro-step, so a name that passes here is final
        for this micro-step once that launch has been issued (pop_written: the overlapped all-reduce's readiness signal)."""
        for n in names:
            if n not in self._written:
                self._written.add(n)
                self._new.append(n)
        if getattr(self, "_touched", None) is None:
            return True
        seen = [n in self._touched for n in names]
        assert all(seen) or not any(seen), names
        self._touched.update(names)
        return seen[0]

    def pop_written(self):
        """Names whose gradient launch has been issued since the last call."""
        out, self._new = self._new, []
        return out

    def finish_step(self):
        if getattr(self, "_touched", None) is None:
            return
        for name in self.spec:
            if name not in self._touched:
                self.g(name).zero_()
        self._touched = None

    def sync_shadow(self):
        """16-bit shadow <- fp32 master (library kernel; after an external optimizer wrote the master)."""
        if self.master.is_cuda:
            L.check(L.lib().dfw_convert_f32(self.master.data_ptr(), self.shadow.data_ptr(), self.numel,
                                            L.BF16 if self.dtype == torch.bfloat16 else L.F16,
                                            torch.cuda.current_stream().cuda_stream), "dfw_convert_f32")
        else:
            self.shadow.copy_(self.master)
        self.version += 1


class UNetTrainer:
    """fwd + bwd (+ optimizer) of MyUNet2DConditionModel in training mode.  Boundary surface the launcher touches
    (T:1105, 1136, 1163, 1189, 1376-1379): train(), parameters(), enable_gradient_checkpointing(),
    clear_attn_bank(), save_pretrained(), state_dict()."""

    def __init__(self, config, state_dict, torch_dtype=torch.bfloat16, device="cuda", loss_scale=1.0,
                 dynamic_loss_scale=None, growth_interval=2000):
        cfg = weights.default_unet_config()
        cfg.update(config or {})
        self.config = cfg
        self.dtype, self.device = torch_dtype, torch.device(device)
        L.lib()
        weights.check_state_dict(state_dict, weights.unet_param_shapes(cfg), "unet")
        self.loss_scale = float(loss_scale)
        # fp16: GradScaler's dynamic scale (halve on overflow, double after growth_interval clean steps); bf16 needs none
        self.dynamic_loss_scale = (torch_dtype == torch.float16) if dynamic_loss_scale is None else bool(dynamic_loss_scale)
        self.growth_interval, self._good_steps, self.skipped_steps = int(growth_interval), 0, 0
        self._overflow_pending = None
        self._param, self._master_seen, self._pending_ref, self.reducer = None, 0, None, None
        self._graphs = {}
        self.graph_recaptures = 0     # captured steps dropped and re-captured because the loss scale changed
        self._cs = ob.ColsumQueue()
        self.training = True
        self.step_count = 0
        self._derived, self._derived_version, self._derived_table = {}, -1, None
        self._build(state_dict)

    # ------------------------------------------------------------------ parameters
    def _resnet_prefixes(self):
        cfg = self.config
        lpb, nb = cfg["layers_per_block"], len(cfg["block_out_channels"])
        out = [f"down_blocks.{i}.resnets.{j}." for i in range(nb) for j in range(lpb)]
        out += ["mid_block.resnets.0.", "mid_block.resnets.1."]
        out += [f"up_blocks.{i}.resnets.{j}." for i in range(nb) for j in range(lpb + 1)]
        return out

    def _transformer_prefixes(self):
        cfg = self.config
        lpb = cfg["layers_per_block"]
        out = []
        for i, typ in enumerate(cfg["down_block_types"]):
            if typ == "CrossAttnDownBlock2D":
                out += [f"down_blocks.{i}.attentions.{j}." for j in range(lpb)]
        for i, typ in enumerate(cfg["up_block_types"]):
            if typ == "CrossAttnUpBlock2D":
                out += [f"up_blocks.{i}.attentions.{j}." for j in range(lpb + 1)]
        out.append("mid_block.attentions.0.")
        return out

    def _build(self, sd):
        cfg = self.config
        real = ParamStore(self.device, self.dtype)
        self.P = real
        f = lambda k: sd[k].float()

        class _Pend:       # collect (name -> tensor) first; the flat layout is laid out in FORWARD order below
            def __init__(self):
                self.items = {}

            def add(self, name, t):
                self.items[name] = t
        P = _Pend()
        P.add("conv_in.weight", packing.pack_conv_small(sd["conv_in.weight"]))
        P.add("conv_in.bias", f("conv_in.bias"))
        P.add("conv_in_ref.weight", packing.pack_conv_small(sd["conv_in_ref.weight"]))
        P.add("conv_in_ref.bias", f("conv_in_ref.bias"))
        for n in ("linear_1", "linear_2"):
            P.add(f"time_embedding.{n}.weight", f(f"time_embedding.{n}.weight"))
            P.add(f"time_embedding.{n}.bias", f(f"time_embedding.{n}.bias"))
        rp = self._resnet_prefixes()
        P.add("tp_w", torch.cat([f(p + "time_emb_proj.weight") for p in rp], 0))
        P.add("tp_b", torch.cat([f(p + "time_emb_proj.bias") for p in rp], 0))
        self.res = {}
        off = 0
        for p in rp:
            cout, cin = sd[p + "conv1.weight"].shape[:2]
            self.res[p] = dict(cin=cin, cout=cout, tslice=(off, cout), short=p + "conv_shortcut.weight" in sd)
            off += cout
            for n in ("norm1", "norm2"):
                P.add(p + n + ".weight", f(p + n + ".weight"))
                P.add(p + n + ".bias", f(p + n + ".bias"))
            for n in ("conv1", "conv2"):
                P.add(p + n + ".weight", packing.pack_conv3x3(f(p + n + ".weight")))
                P.add(p + n + ".bias", f(p + n + ".bias"))
            if self.res[p]["short"]:
                P.add(p + "conv_shortcut.weight", packing.pack_conv1x1(f(p + "conv_shortcut.weight")))
                P.add(p + "conv_shortcut.bias", f(p + "conv_shortcut.bias"))
        self.tp_total = off
        tp = self._transformer_prefixes()
        heads = cfg["attention_head_dim"]
        boc = list(cfg["block_out_channels"])
        heads = [heads] * len(boc) if isinstance(heads, int) else list(heads)
        self.tr = {}
        kv, koff = [], 0
        for p in tp:
            b = p + "transformer_blocks.0."
            C = sd[p + "proj_in.weight"].shape[0]
            self.tr[p] = dict(C=C, heads=C // 64, kvslice=(koff, 2 * C))
            koff += 2 * C
            kv.append(torch.cat([f(b + "attn2.to_k.weight"), f(b + "attn2.to_v.weight")], 0))
            P.add(p + "norm.weight", f(p + "norm.weight")); P.add(p + "norm.bias", f(p + "norm.bias"))
            P.add(p + "proj_in.weight", f(p + "proj_in.weight")); P.add(p + "proj_in.bias", f(p + "proj_in.bias"))
            for n in ("norm1", "norm2", "norm3"):
                P.add(b + n + ".weight", f(b + n + ".weight")); P.add(b + n + ".bias", f(b + n + ".bias"))
            P.add(b + "attn1.qkv", torch.cat([f(b + "attn1.to_q.weight"), f(b + "attn1.to_k.weight"), f(b + "attn1.to_v.weight")], 0))
            P.add(b + "attn1.to_out.0.weight", f(b + "attn1.to_out.0.weight")); P.add(b + "attn1.to_out.0.bias", f(b + "attn1.to_out.0.bias"))
            P.add(b + "attn2.to_q.weight", f(b + "attn2.to_q.weight"))
            P.add(b + "attn2.to_out.0.weight", f(b + "attn2.to_out.0.weight")); P.add(b + "attn2.to_out.0.bias", f(b + "attn2.to_out.0.bias"))
            wp, bp = packing.pack_geglu(f(b + "ff.net.0.proj.weight"), f(b + "ff.net.0.proj.bias"))
            P.add(b + "ff1.weight", wp); P.add(b + "ff1.bias", bp)
            P.add(b + "ff.net.2.weight", f(b + "ff.net.2.weight")); P.add(b + "ff.net.2.bias", f(b + "ff.net.2.bias"))
            P.add(p + "proj_out.weight", f(p + "proj_out.weight")); P.add(p + "proj_out.bias", f(p + "proj_out.bias"))
        self.kv_total = koff
        P.add("kv_w_all", torch.cat(kv, 0))
        self.samplers = []
        for i in range(len(boc)):
            for kind, name in (("down", f"down_blocks.{i}.downsamplers.0.conv."), ("up", f"up_blocks.{i}.upsamplers.0.conv.")):
                if name + "weight" in sd:
                    P.add(name + "weight", packing.pack_conv3x3(f(name + "weight")))
                    P.add(name + "bias", f(name + "bias"))
                    self.samplers.append(name)
        P.add("conv_norm_out.weight", f("conv_norm_out.weight")); P.add("conv_norm_out.bias", f("conv_norm_out.bias"))
        # conv_out: 4 output channels padded to 8 rows (rows 4..7 stay zero: zero weights, zero gradients), so the
        # weight gradient of the padded output gradient lands in place
        wo = packing.pack_conv3x3(f("conv_out.weight"))
        P.add("conv_out.weight", torch.cat([wo, torch.zeros(8 - wo.shape[0], wo.shape[1])], 0))
        P.add("conv_out.bias", torch.cat([f("conv_out.bias"), torch.zeros(8 - wo.shape[0])], 0))
        # Flat layout = the order the forward USES the parameters, so the backward finishes their gradients from the END of
        # the buffer towards its start: contiguous buckets complete one after the other and the gradient all-reduce of a
        # bucket is issued while the backward of the earlier layers still runs (GradBucketReducer; DDP's overlap, T:1226-1228).
        nb, lpb = len(boc), cfg["layers_per_block"]
        order = ["time_embedding.linear_1.", "time_embedding.linear_2.", "tp_w", "tp_b", "kv_w_all", "conv_in_ref.", "conv_in."]
        for i, typ in enumerate(cfg["down_block_types"]):
            for j in range(lpb):
                order.append(f"down_blocks.{i}.resnets.{j}.")
                if typ == "CrossAttnDownBlock2D":
                    order.append(f"down_blocks.{i}.attentions.{j}.")
            order.append(f"down_blocks.{i}.downsamplers.0.conv.")
        order += ["mid_block.resnets.0.", "mid_block.attentions.0.", "mid_block.resnets.1."]
        for i, typ in enumerate(cfg["up_block_types"]):
            for j in range(lpb + 1):
                order.append(f"up_blocks.{i}.resnets.{j}.")
                if typ == "CrossAttnUpBlock2D":
                    order.append(f"up_blocks.{i}.attentions.{j}.")
            order.append(f"up_blocks.{i}.upsamplers.0.conv.")
        order += ["conv_norm_out.", "conv_out."]
        left = dict(P.items)
        for pre in order:
            for name in [n for n in left if n.startswith(pre)]:
                real.add(name, left.pop(name))
        assert not left, sorted(left)
        P = real
        P.finalize()
        self.groups, self.eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.heads_by_level = heads
        if P.master.is_cuda:
            self._found_inf = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._found_host = torch.zeros(1, dtype=torch.int32).pin_memory()

    # ------------------------------------------------------------------ boundary surface of the training launcher
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def requires_grad_(self, flag=True):
        return self

    def enable_gradient_checkpointing(self):
        """T:1163.  Not needed on 288 GB: the forward keeps its 16-bit activations (~2 GB for a 7-shot episode)."""
        return None

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None

    def clear_attn_bank(self):
        """T:1376-1379: the lock-step pass holds no bank state between calls."""
        return None

    @property
    def module(self):       # `unet.module.clear_attn_bank()` under DDP (T:1377)
        return self

    def parameters(self):
        """ONE torch.nn.Parameter over the flat fp32 master buffer (packed layout): what `torch.optim.AdamW(unet.parameters())`
        (T:1186-1194), `clip_grad_norm_(unet.parameters(), ...)` (T:1393) and the gradient all-reduce operate on.  Its
        `.grad` is the flat fp32 gradient (a view of P.grad) after a backward, None after optimizer.zero_grad().  An
        in-place update by an external optimizer is noticed through the tensor's version counter and the 16-bit shadow
        the kernels read is refreshed before the next forward."""
        if self._param is None:
            self._param = torch.nn.Parameter(self.P.master, requires_grad=True)
            self._master_seen = self._param._version
        return [self._param]

    def export(self, flat):
        """flat packed buffer (master or grad) -> dict in the diffusers key layout."""
        P, out = self.P, {}
        v = lambda name: P._view(flat, name)
        out["conv_in.weight"] = v("conv_in.weight").reshape(-1, 3, 3, self.config["in_channels"]).permute(0, 3, 1, 2).contiguous()
        out["conv_in.bias"] = v("conv_in.bias").clone()
        out["conv_in_ref.weight"] = v("conv_in_ref.weight").reshape(-1, 3, 3, self.config["in_channels_ref"]).permute(0, 3, 1, 2).contiguous()
        out["conv_in_ref.bias"] = v("conv_in_ref.bias").clone()
        for n in ("linear_1", "linear_2"):
            out[f"time_embedding.{n}.weight"] = v(f"time_embedding.{n}.weight").clone()
            out[f"time_embedding.{n}.bias"] = v(f"time_embedding.{n}.bias").clone()
        tpw, tpb = v("tp_w"), v("tp_b")
        for p, r in self.res.items():
            o, c = r["tslice"]
            out[p + "time_emb_proj.weight"], out[p + "time_emb_proj.bias"] = tpw[o:o + c].clone(), tpb[o:o + c].clone()
            for n in ("norm1", "norm2"):
                out[p + n + ".weight"], out[p + n + ".bias"] = v(p + n + ".weight").clone(), v(p + n + ".bias").clone()
            for n in ("conv1", "conv2"):
                w = v(p + n + ".weight")
                out[p + n + ".weight"] = packing.unpack_conv3x3_grad(w.view(w.shape[0], 9, -1))
                out[p + n + ".bias"] = v(p + n + ".bias").clone()
            if r["short"]:
                w = v(p + "conv_shortcut.weight")
                out[p + "conv_shortcut.weight"] = w.reshape(w.shape[0], w.shape[1], 1, 1).clone()
                out[p + "conv_shortcut.bias"] = v(p + "conv_shortcut.bias").clone()
        kvw = v("kv_w_all")
        for p, t in self.tr.items():
            b, C = p + "transformer_blocks.0.", t["C"]
            for n in ("norm.weight", "norm.bias", "proj_in.weight", "proj_in.bias", "proj_out.weight", "proj_out.bias"):
                out[p + n] = v(p + n).clone()
            for n in ("norm1", "norm2", "norm3"):
                out[b + n + ".weight"], out[b + n + ".bias"] = v(b + n + ".weight").clone(), v(b + n + ".bias").clone()
            qkv = v(b + "attn1.qkv")
            out[b + "attn1.to_q.weight"], out[b + "attn1.to_k.weight"], out[b + "attn1.to_v.weight"] = (qkv[i * C:(i + 1) * C].clone() for i in range(3))
            for n in ("attn1.to_out.0.weight", "attn1.to_out.0.bias", "attn2.to_q.weight", "attn2.to_out.0.weight", "attn2.to_out.0.bias",
                      "ff.net.2.weight", "ff.net.2.bias"):
                out[b + n] = v(b + n).clone()
            o, _ = t["kvslice"]
            out[b + "attn2.to_k.weight"], out[b + "attn2.to_v.weight"] = kvw[o:o + C].clone(), kvw[o + C:o + 2 * C].clone()
            w1, b1 = v(b + "ff1.weight"), v(b + "ff1.bias")
            inv = torch.argsort(packing.geglu_perm(w1.shape[0] // 2)).to(w1.device)
            out[b + "ff.net.0.proj.weight"], out[b + "ff.net.0.proj.bias"] = w1[inv].contiguous(), b1[inv].contiguous()
        for name in self.samplers:
            w = v(name + "weight")
            out[name + "weight"] = packing.unpack_conv3x3_grad(w.view(w.shape[0], 9, -1))
            out[name + "bias"] = v(name + "bias").clone()
        out["conv_norm_out.weight"], out["conv_norm_out.bias"] = v("conv_norm_out.weight").clone(), v("conv_norm_out.bias").clone()
        oc = self.config["out_channels"]
        w = v("conv_out.weight")[:oc]
        out["conv_out.weight"] = packing.unpack_conv3x3_grad(w.reshape(oc, 9, -1))
        out["conv_out.bias"] = v("conv_out.bias")[:oc].clone()
        return out

    def state_dict(self):
        return self.export(self.P.master)

    def grad_dict(self):
        return self.export(self.P.grad)

    def save_pretrained(self, path, subfolder=None):
        weights.save_pretrained(path, dict(self.config), {k: v.cpu() for k, v in self.state_dict().items()}, subfolder)

    # ------------------------------------------------------------------ derived weight layouts (once per step)
    def _d(self, kind, name):
        """16-bit copies the data-gradient GEMMs read: 'T' = W^T of a Linear ([K, N] as the forward kernel's [N'][K']),
        'D' = tap-mirrored, channel-swapped conv3x3 weight.  Rebuilt lazily after every optimizer step."""
        if self._derived_version != self.P.version:
            if self._derived_table is not None:
                ob.weight_relayout_batch(self._derived_table)      # one launch rewrites every copy in place
            else:
                self._derived = {}
            self._derived_version = self.P.version
        key = (kind, name)
        if key not in self._derived:                               # first step (or a new consumer): build lazily
            w = self.P.w(name)
            self._derived[key] = ob.linear_wt(w) if kind == "T" else ob.conv3x3_wd(w, w.shape[0])
            self._derived_table = None
        return self._derived[key]

    def _freeze_derived(self):
        """After a backward: the set of derived copies is known -- refresh them with one batched launch from now on."""
        if self._derived_table is None and self._derived:
            entries = [(kind, self.P.w(name), y) for (kind, name), y in self._derived.items()]
            self._derived_table = ob.relayout_table(entries, self.device)

    # ------------------------------------------------------------------ ops with registered backward
    def _linear(self, tape, x, wname, bname=None, residual=None, colscale=None, need_dx=True):
        P, gs = self.P, 1.0 / self.loss_scale
        y = ops.linear(x, P.w(wname), bias=P.p(bname) if bname else None, residual=residual, colscale=colscale)

        def bwd(dy):
            N, K = P.spec[wname][1]
            ob.gemm_tn(dy, x, out=P.g(wname).view(1, N, 1, K), accumulate=P.acc(wname), scale=gs)
            if bname:
                self._cs.add(dy, P.g(bname).view(1, N), accumulate=P.acc(bname), scale=gs)
            if need_dx:
                tape.accum(x, ops.linear(dy, self._d("T", wname)))
            if residual is not None:
                tape.accum(residual, dy)
        tape.add(y, bwd)
        return y

    def _conv(self, tape, x, wname, bname, stride=1, ups=False, rowbias=None, residual=None, dtproj=None, tslice=None,
              need_dx=True):
        P, gs = self.P, 1.0 / self.loss_scale
        cout = P.spec[wname][1][0]
        B, Hi, Wi, Cin = x.shape
        y = ops.conv3x3(x, P.w(wname), cout, bias=P.p(bname), stride=stride, pad=1, ups=ups, rowbias=rowbias, residual=residual)
        Ho, Wo = y.shape[1:3]

        def bwd(dy):
            ob.gemm_tn(dy, x, taps=9, geom=(Hi, Wi, Ho, Wo, stride, 1, int(ups)), out=P.g(wname).view(1, cout, 9, Cin),
                       accumulate=P.acc(wname), scale=gs)
            self._cs.add(dy, P.g(bname).view(1, cout), accumulate=P.acc(bname), scale=gs)
            if tslice is not None:      # d(time_emb_proj output)[img] = sum over the image's pixels (kept at loss scale)
                o, c = tslice
                self._cs.add(dy, dtproj[:, o:o + c], segs=B)
            if need_dx:
                wd = self._d("D", wname)
                if stride == 2:
                    dx = ops.conv3x3(ob.zero_stuff2x(dy), wd, Cin)
                elif ups:
                    dx = ob.pool2x2_sum(ops.conv3x3(dy, wd, Cin))
                else:
                    dx = ops.conv3x3(dy, wd, Cin)
                tape.accum(x, dx)
            if residual is not None:
                tape.accum(residual, dy)
        tape.add(y, bwd)
        return y

    def _gn(self, tape, x, gname, bname, eps, silu):
        P, gs = self.P, 1.0 / self.loss_scale
        y, mr = ops.groupnorm(x, P.p(gname), P.p(bname), self.groups, eps, silu=silu, return_stats=True)

        def bwd(dy):
            # x's other consumer (the block's residual add, a skip concat) has run already: its gradient rides into the kernel
            tape.accum(x, ob.groupnorm_bwd(x, dy, mr, P.p(gname), P.p(bname), self.groups, silu, P.g(gname), P.g(bname),
                                           accumulate=P.acc(gname, bname), grad_scale=gs, dx_add=tape.take(x)))
        tape.add(y, bwd)
        return y

    def _ln(self, tape, x, gname, bname):
        P, gs = self.P, 1.0 / self.loss_scale
        y = ops.layernorm(x, P.p(gname), P.p(bname))

        def bwd(dy):
            tape.accum(x, ob.layernorm_bwd(x, dy, P.p(gname), P.g(gname), P.g(bname), accumulate=P.acc(gname, bname), grad_scale=gs,
                                           dx_add=tape.take(x)))
        tape.add(y, bwd)
        return y

    def _resnet(self, tape, p, x, tproj, dtproj):
        r = self.res[p]
        B, H, W, Cin = x.shape
        h = self._gn(tape, x, p + "norm1.weight", p + "norm1.bias", self.eps, True)
        o, c = r["tslice"]
        h = self._conv(tape, h, p + "conv1.weight", p + "conv1.bias", rowbias=tproj[:, o:o + c], dtproj=dtproj, tslice=r["tslice"])
        h = self._gn(tape, h, p + "norm2.weight", p + "norm2.bias", self.eps, True)
        sc = x
        if r["short"]:
            sc = self._linear(tape, x.view(-1, Cin), p + "conv_shortcut.weight", p + "conv_shortcut.bias").view(B, H, W, r["cout"])
        return self._conv(tape, h, p + "conv2.weight", p + "conv2.bias", residual=sc)

    def _transformer(self, tape, p, x, kv_all, dkv_all, L_ctx, n_ref):
        P = self.P
        t = self.tr[p]
        B, H, W, C = x.shape
        N, heads = H * W, t["heads"]
        b = p + "transformer_blocks.0."
        n = self._gn(tape, x, p + "norm.weight", p + "norm.bias", 1e-6, False)
        t0 = self._linear(tape, n.view(-1, C), p + "proj_in.weight", p + "proj_in.bias")
        l1 = self._ln(tape, t0, b + "norm1.weight", b + "norm1.bias")
        qkv = self._linear(tape, l1, b + "attn1.qkv", colscale=(C, ops.FSA_QSCALE))
        q3 = qkv.view(B, N, 3 * C)
        q, k, v = q3[..., :C], q3[..., C:2 * C], q3[..., 2 * C:]
        lse = torch.empty(B, heads, N, dtype=torch.float32, device=x.device)
        nshot = n_ref // (B - n_ref) if n_ref else 0
        if n_ref:
            att = ops.fsa_attention(q, k, v, heads, k[:n_ref], v[:n_ref], nshot=nshot, n_plain=n_ref, q_prescaled=True, lse=lse)
        else:
            att = ops.fsa_attention(q, k, v, heads, q_prescaled=True, lse=lse)

        def att_bwd(datt):
            tape.accum(qkv, ob.fsa_attention_bwd(q3, att, datt, lse, heads, nshot=nshot, n_plain=n_ref).view(-1, 3 * C))
        tape.add(att, att_bwd)
        t1 = self._linear(tape, att.view(-1, C), b + "attn1.to_out.0.weight", b + "attn1.to_out.0.bias", residual=t0)
        l2 = self._ln(tape, t1, b + "norm2.weight", b + "norm2.bias")
        # attn2 on the MFMA path: the prompt's 77 keys are two 64-key tiles of the flash kernels (ragged one masked); q
        # leaves to_q pre-scaled like attn1's.  (dfw_cross_attention / _bwd, the VALU kernels of the folded-prompt
        # inference path, took 6.7 ms of the 7-shot step here.)
        q2 = self._linear(tape, l2, b + "attn2.to_q.weight", colscale=(C, ops.FSA_QSCALE))
        o, _ = t["kvslice"]
        kv2 = kv_all.view(B, L_ctx, -1)[..., o:o + 2 * C]
        q2v = q2.view(B, N, C)
        lse2 = torch.empty(B, heads, N, dtype=torch.float32, device=x.device)
        ca = ops.fsa_attention(q2v, kv2[..., :C], kv2[..., C:], heads, q_prescaled=True, lse=lse2)

        def ca_bwd(dca):
            dkv = dkv_all.view(B, L_ctx, -1)[..., o:o + 2 * C]
            tape.accum(q2, ob.attention_bwd(q2v, kv2[..., :C], kv2[..., C:], ca, dca.contiguous(), lse2, heads,
                                            dkv[..., :C], dkv[..., C:]).view(-1, C))
        tape.add(ca, ca_bwd)
        t2 = self._linear(tape, ca.view(-1, C), b + "attn2.to_out.0.weight", b + "attn2.to_out.0.bias", residual=t1)
        l3 = self._ln(tape, t2, b + "norm3.weight", b + "norm3.bias")
        pre = self._linear(tape, l3, b + "ff1.weight", b + "ff1.bias")
        ff = ob.geglu_fwd(pre)
        tape.add(ff, lambda dff: tape.accum(pre, ob.geglu_bwd(pre, dff)))
        t3 = self._linear(tape, ff, b + "ff.net.2.weight", b + "ff.net.2.bias", residual=t2)
        out = self._linear(tape, t3, p + "proj_out.weight", p + "proj_out.bias", residual=x.view(-1, C))
        return out.view(B, H, W, C)

    # ------------------------------------------------------------------ the step
    def _timestep_rows(self, timestep, Bt):
        """[Bt] fp32 device vector of the step's timestep (T:1365 passes a cuda LongTensor [bsz]; no host sync here)."""
        if torch.is_tensor(timestep):
            return timestep.to(device=self.device, dtype=torch.float32).reshape(-1)[:1].expand(Bt).contiguous()
        return torch.full((Bt,), float(timestep), dtype=torch.float32, device=self.device)

    def _forward(self, z_refcat, z_tag, timestep, ehs, zero_grad=True, ehs_ref=None):
        """Lock-step forward over [support ; query] latents with the backward tape.  Returns the context _backward needs;
        ctx['pred'] is the query rows' prediction [b, 4, h, w] fp32 (the support pass' output is discarded, T:1381)."""
        P, dt, dev, cfg = self.P, self.dtype, self.device, self.config
        if self.dynamic_loss_scale:      # the scale this step's kernels use depends on the previous step's overflow flag
            self._resolve_overflow()
        if self._param is not None and self._param._version != self._master_seen:
            P.sync_shadow()              # an external optimizer (torch.optim.AdamW on parameters()) rewrote the master in place
            self._master_seen = self._param._version
        tape = _Tape()
        gs = 1.0 / self.loss_scale
        P.begin_step(fresh=zero_grad)
        zr = z_refcat.to(dev, torch.float32).contiguous()
        zq = z_tag.to(dev, torch.float32).contiguous()
        n_ref, bq = zr.shape[0], zq.shape[0]
        Bt = n_ref + bq
        c0 = cfg["block_out_channels"][0]
        # ---- time embedding (U:991-1015) and the 22 fused time projections
        t = self._timestep_rows(timestep, Bt)
        temb = ops.timestep_embedding(t, c0, dt, cfg["flip_sin_to_cos"], float(cfg["freq_shift"]))
        e1 = self._linear(tape, temb, "time_embedding.linear_1.weight", "time_embedding.linear_1.bias", need_dx=False)
        a1 = ob.silu(e1)
        tape.add(a1, lambda d: tape.accum(e1, ob.silu(e1, d)))
        e2 = self._linear(tape, a1, "time_embedding.linear_2.weight", "time_embedding.linear_2.bias")
        a2 = ob.silu(e2)
        tape.add(a2, lambda d: tape.accum(e2, ob.silu(e2, d)))
        tproj = ops.linear(a2, P.w("tp_w"), bias=P.p("tp_b"), out_f32=True)                  # [Bt, sum Cout] fp32
        dtproj = ops.zeros((Bt, self.tp_total), torch.float32, dev)                         # filled by the resnets' closures

        def tproj_bwd(_):
            # every resnet has queued its column slice of dtproj by now (they sit later on the tape): issue the queue
            self._cs.flush()
            d16 = ob.nchw_to_nhwc(dtproj.view(Bt, self.tp_total, 1, 1), dt, cp=self.tp_total).view(Bt, self.tp_total)
            ob.gemm_tn(d16, a2, out=P.g("tp_w").view(1, self.tp_total, 1, a2.shape[1]), accumulate=P.acc("tp_w"), scale=gs)
            self._cs.add(d16, P.g("tp_b").view(1, self.tp_total), accumulate=P.acc("tp_b"), scale=gs)
            tape.accum(a2, ops.linear(d16, self._d("T", "tp_w")))
        tape.add(tproj, tproj_bwd)
        # ---- prompt K/V of all layers in one GEMM
        e = ehs.to(dev, dt)
        if e.shape[0] == 1:
            e = e.expand(bq, -1, -1)
        if ehs_ref is not None:          # the launcher's own rows for the support images (T:1369: the same prompt, repeated)
            er = ehs_ref.to(dev, dt)
            if er.shape[0] == 1:
                er = er.expand(n_ref, -1, -1)
            e = torch.cat([er, e], 0) if n_ref else e
        elif e.shape[0] != Bt:
            e = e[:1].expand(Bt, -1, -1)
        if e.shape[0] != Bt:
            raise ValueError(f"encoder_hidden_states rows {e.shape[0]} do not match the {Bt} latents of the step")
        L_ctx = e.shape[1]
        ehs2d = e.reshape(Bt * L_ctx, e.shape[2]).contiguous()
        kv_all = ops.linear(ehs2d, P.w("kv_w_all"))
        dkv_all = ops.zeros(kv_all.shape, kv_all.dtype, dev)                                 # filled by the attn2 closures
        tape.add(kv_all, lambda _: ob.gemm_tn(dkv_all, ehs2d, out=P.g("kv_w_all").view(1, self.kv_total, 1, ehs2d.shape[1]),
                                              accumulate=P.acc("kv_w_all"), scale=gs))
        # ---- conv_in_ref | conv_in (U:1117-1121)
        h, w = zq.shape[2:]
        x = torch.empty(Bt, h, w, c0, dtype=dt, device=dev)
        if n_ref:
            ops.conv_small(zr, P.p("conv_in_ref.weight"), P.p("conv_in_ref.bias"), c0, 9, dt, out=x[:n_ref])
        ops.conv_small(zq, P.p("conv_in.weight"), P.p("conv_in.bias"), c0, 9, dt, out=x[n_ref:])

        def conv_in_bwd(dx):
            geom = (h, w, h, w, 1, 1, 0)
            if n_ref:
                zin = ob.nchw_to_nhwc(zr, dt, cp=8)
                ob.gemm_tn(dx[:n_ref], zin, taps=9, geom=geom, out=P.g("conv_in_ref.weight").view(1, c0, 9, 8), accumulate=P.acc("conv_in_ref.weight"), scale=gs)
                self._cs.add(dx[:n_ref], P.g("conv_in_ref.bias").view(1, c0), accumulate=P.acc("conv_in_ref.bias"), scale=gs)
            zin = ob.nchw_to_nhwc(zq, dt, cp=8)
            g8 = ob.gemm_tn(dx[n_ref:], zin, taps=9, geom=geom, scale=gs)                  # [1, c0, 9, 8]: 4 real input channels
            if P.acc("conv_in.weight"):
                P.g("conv_in.weight").add_(g8.view(c0, 9, 8)[..., :cfg["in_channels"]])
            else:
                P.g("conv_in.weight").copy_(g8.view(c0, 9, 8)[..., :cfg["in_channels"]])
            self._cs.add(dx[n_ref:], P.g("conv_in.bias").view(1, c0), accumulate=P.acc("conv_in.bias"), scale=gs)
        tape.add(x, conv_in_bwd)
        # ---- trunk (U:1153-1243)
        lpb, nb = cfg["layers_per_block"], len(cfg["block_out_channels"])
        skips = [x]
        for i, typ in enumerate(cfg["down_block_types"]):
            for j in range(lpb):
                x = self._resnet(tape, f"down_blocks.{i}.resnets.{j}.", x, tproj, dtproj)
                if typ == "CrossAttnDownBlock2D":
                    x = self._transformer(tape, f"down_blocks.{i}.attentions.{j}.", x, kv_all, dkv_all, L_ctx, n_ref)
                skips.append(x)
            if i != nb - 1:
                x = self._conv(tape, x, f"down_blocks.{i}.downsamplers.0.conv.weight", f"down_blocks.{i}.downsamplers.0.conv.bias", stride=2)
                skips.append(x)
        x = self._resnet(tape, "mid_block.resnets.0.", x, tproj, dtproj)
        x = self._transformer(tape, "mid_block.attentions.0.", x, kv_all, dkv_all, L_ctx, n_ref)
        x = self._resnet(tape, "mid_block.resnets.1.", x, tproj, dtproj)
        for i, typ in enumerate(cfg["up_block_types"]):
            for j in range(lpb + 1):
                s = skips.pop()
                cat = ops.concat_channels(x, s)
                ca_, cs_ = x.shape[-1], s.shape[-1]

                def cat_bwd(d, x=x, s=s, ca_=ca_, cs_=cs_):
                    tape.accum(x, ob.slice_channels(d, 0, ca_))
                    tape.accum(s, ob.slice_channels(d, ca_, cs_))
                tape.add(cat, cat_bwd)
                x = self._resnet(tape, f"up_blocks.{i}.resnets.{j}.", cat, tproj, dtproj)
                if typ == "CrossAttnUpBlock2D":
                    x = self._transformer(tape, f"up_blocks.{i}.attentions.{j}.", x, kv_all, dkv_all, L_ctx, n_ref)
            if i != nb - 1:
                x = self._conv(tape, x, f"up_blocks.{i}.upsamplers.0.conv.weight", f"up_blocks.{i}.upsamplers.0.conv.bias", ups=True)
        # ---- out (U:1246-1249)
        hn = self._gn(tape, x, "conv_norm_out.weight", "conv_norm_out.bias", self.eps, True)
        oc = cfg["out_channels"]
        pred_all = ops.conv3x3(hn, P.w("conv_out.weight")[:oc], oc, bias=P.p("conv_out.bias")[:oc], out_nchw_f32=True)
        # gradient seeds of conv_out: support rows stay zero (pred_ref * 0, T:1381); zeroed by a kernel (capture-safe)
        dpred = ops.zeros((Bt, h, w, 8), dt, dev)
        dpn = ops.zeros((Bt, oc, h, w), torch.float32, dev)
        return dict(tape=tape, hn=hn, pred_all=pred_all, pred=pred_all[n_ref:], n_ref=n_ref, h=h, w=w, c0=c0, oc=oc,
                    tproj=tproj, dtproj=dtproj, kv_all=kv_all, dkv_all=dkv_all, dpred=dpred, dpn=dpn)

    def _backward(self, c, reducer=None):
        """Backward of _forward's graph from the loss gradient already deposited in c['dpred'] (NHWC, 8 channels, storage
        dtype, times the loss scale) / c['dpn'] (the same rounded values, NCHW fp32).  Parameter gradients land in P.grad.
        reducer: a GradBucketReducer -- told after every closure which parameter gradients have just been issued, so a
        bucket's all-reduce starts as soon as its last writer is in the stream."""
        P, dt, gs = self.P, self.dtype, 1.0 / self.loss_scale
        tape, hn, h, w, c0, oc = c["tape"], c["hn"], c["h"], c["w"], c["c0"], c["oc"]
        # bias / time-projection column sums are queued and issued in batches (ob.ColsumQueue): with a reducer, before every
        # readiness report (a gradient is final only once its launch is in the stream); otherwise twice per walk
        # (round 4: the queue is flushed only when the names reported so far would COMPLETE a bucket -- flushing after every
        # closure issued the ~190 column sums one by one and cost the reducer path 1.5 ms per step against the fused one)
        pend = []

        def _note(final=False):
            pend.extend(P.pop_written())
            if pend and (final or reducer.would_fire(pend)):
                self._cs.flush()
                reducer.mark(list(pend))
                pend.clear()
        note = _note if reducer is not None else None
        # conv_out backward: weight / bias gradients in place (padded to 8 rows), data gradient by the direct conv
        ob.gemm_tn(c["dpred"], hn, taps=9, geom=(h, w, h, w, 1, 1, 0), out=P.g("conv_out.weight").view(1, 8, 9, c0), accumulate=P.acc("conv_out.weight"), scale=gs)
        self._cs.add(c["dpred"], P.g("conv_out.bias").view(1, 8), accumulate=P.acc("conv_out.bias"), scale=gs)
        wdo = P.p("conv_out.weight")[:oc].view(oc, 9, c0).flip(1).permute(2, 1, 0).contiguous()     # [c0][tap'][oc] fp32
        dhn = ops.conv_small(c["dpn"], wdo, None, c0, 9, dt)
        if note:
            note()
        tape.accum(c["tproj"], c["dtproj"])   # seeds of the two conditioning paths: their buffers fill up during the walk
        tape.accum(c["kv_all"], c["dkv_all"])
        tape.backward(hn, dhn, after=note)
        self._cs.flush()
        P.finish_step()
        if note:
            note(final=True)
        self._freeze_derived()

    def forward_backward(self, z_refcat, z_tag, target, timestep, ehs, zero_grad=True, reducer=None):
        """One micro-step: lock-step forward over [support ; query] latents, MSE(pred, target), backward.
        z_refcat [b*s, 8, h, w] (cat([z_ref, z_mask_ref], 1), T:1360-1362), z_tag [b, 4, h, w], target [b, 4, h, w]
        (= -z_mask_tag, T:1371), all fp32 NCHW; ehs [1, L, D] or [b, L, D] prompt embedding (T:1368).
        Returns (loss fp32 tensor [1], pred [b, 4, h, w] fp32).  Gradients accumulate into P.grad (fp32, packed).
        reducer (GradBucketReducer): all-reduce the gradient buckets over the ranks WHILE the backward runs; the loss is
        averaged with the last bucket (reducer.finish() returns it)."""
        c = self._forward(z_refcat, z_tag, timestep, ehs, zero_grad)
        n_ref = c["n_ref"]
        tgt = target.to(self.device, torch.float32).contiguous()
        loss, _ = ob.mse_loss(c["pred"], tgt, self.dtype, loss_scale=self.loss_scale, dpred_out=c["dpred"][n_ref:],
                              dpred_nchw_out=c["dpn"][n_ref:])
        if reducer is not None:
            reducer.begin(loss)
        self._backward(c, reducer)
        return loss, c["pred"]

    def forward_backward_captured(self, z_refcat, z_tag, target, timestep, ehs):
        """forward_backward(zero_grad=True) replayed as ONE HIP graph (~2 200 kernel nodes), captured on first use per input
        shape into static input buffers; returns the graph's (loss, pred) buffers (overwritten by the next call).  The
        optimizer step stays outside (its step count / learning rate are host arguments), and so does an overlapped gradient
        all-reduce: multi-GPU runs use forward_backward(reducer=...) eagerly.  Host-side bookkeeping of the step (first-touch
        flags, derived-weight table) is identical for every step, which is what makes the capture valid."""
        self._resolve_overflow()
        ins = [t.to(self.device, torch.float32).contiguous() for t in (z_refcat, z_tag, target)] + [ehs.to(self.device).contiguous()]
        # The loss scale is baked into the captured launches as a host argument, so a graph serves ONE scale.  It is not part
        # of the cache key: under a dynamic scale (fp16 default) every halving / doubling would otherwise leave a ~2 200-node
        # graph with its private activation pool behind (up to 17 scales, HBM growing until the allocator gives up).  A scale
        # change drops the stale graph -- its pool goes back to the allocator -- and re-captures at the new scale.
        key = (tuple(tuple(t.shape) for t in ins), float(timestep))
        ent = self._graphs.get(key)
        if ent is not None and ent[4] != self.loss_scale:
            del self._graphs[key]
            ent[0].reset()
            ent = None
            self.graph_recaptures += 1
            torch.cuda.empty_cache()
        if ent is None:
            static = [t.clone() for t in ins]
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                    # lazy derived-weight copies, then the frozen batched re-layout table
                    self.forward_backward(static[0], static[1], static[2], float(timestep), static[3])
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph(keep_graph=True)
            # the batched refresh of the derived weight copies (W^T, tap-mirrored conv weights) must be PART of the graph: the
            # optimizer rewrites the 16-bit shadow between replays.  Marking the copies stale makes the first _d() of the
            # captured backward issue it, exactly where an eager step after an optimizer step issues it.
            self._derived_version = -2
            dyn, self.dynamic_loss_scale = self.dynamic_loss_scale, False   # no event wait inside the capture
            try:
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    loss, pred = self.forward_backward(static[0], static[1], static[2], float(timestep), static[3])
            finally:
                self.dynamic_loss_scale = dyn
            from .pipeline import _assert_no_memset_nodes
            self.graph_nodes = _assert_no_memset_nodes(graph)
            graph.instantiate()
            ent = (graph, static, loss, pred, self.loss_scale)
            self._graphs[key] = ent
        graph, static, loss, pred, _ = ent
        for dst, src in zip(static, ins):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        graph.replay()
        return loss, pred

    def forward_backward_segmented(self, z_refcat, z_tag, target, timestep, ehs, reducer):
        """Multi-GPU form of forward_backward_captured (round 4): forward + backward replayed as a CHAIN of HIP graphs cut
        where a gradient bucket becomes final, so that the overlapped all-reduce keeps its place -- after each segment the
        reducer issues that segment's buckets from its comm stream while the next segment already runs.  The eager walk
        needed ~2 200 host-side launches per step for the sake of those readiness marks; here the host issues one graph
        launch per segment (17 for the 866 M-parameter UNet at the default 216 MB buckets).  Captured on first use per input
        shape (and per loss scale, like the monolithic graph); returns (rank-averaged loss [1], pred)."""
        self._resolve_overflow()
        ins = [t.to(self.device, torch.float32).contiguous() for t in (z_refcat, z_tag, target)] + [ehs.to(self.device).contiguous()]
        key = ("segmented", tuple(tuple(t.shape) for t in ins), float(timestep), id(reducer))
        ent = self._graphs.get(key)
        if ent is not None and ent[4] != self.loss_scale:
            del self._graphs[key]
            for g, _ in ent[0]:
                g.reset()
            ent = None
            self.graph_recaptures += 1
            torch.cuda.empty_cache()
        if ent is None:
            static = [t.clone() for t in ins]
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                    # lazy derived-weight copies, then the frozen batched re-layout table
                    self.forward_backward(static[0], static[1], static[2], float(timestep), static[3])
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            self._derived_version = -2                # the refresh of the derived weight copies belongs to the first segment
            dyn, self.dynamic_loss_scale = self.dynamic_loss_scale, False
            rec = _SegmentRecorder(reducer, torch.cuda.graph_pool_handle())
            try:
                with torch.cuda.stream(side):
                    rec._open()
                    loss, pred = self.forward_backward(static[0], static[1], static[2], float(timestep), static[3], reducer=rec)
                    rec.finish()
            finally:
                self.dynamic_loss_scale = dyn
            cur.wait_stream(side)
            from .pipeline import _assert_no_memset_nodes
            self.graph_nodes = 0
            for g, _ in rec.segments:
                self.graph_nodes += _assert_no_memset_nodes(g)
                g.instantiate()
            ent = (rec.segments, static, loss, pred, self.loss_scale)
            self._graphs[key] = ent
        segments, static, loss, pred, _ = ent
        for dst, src in zip(static, ins):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        avg = replay_segments(segments, reducer)
        return avg, pred

    # ------------------------------------------------------------------ the reference's call surface (T:1374-1396)
    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, sample, timestep, encoder_hidden_states, is_target=True, return_dict=True, **unused):
        """MyUNet2DConditionModel.forward under autograd, as the training launcher drives it:
            pred_ref = unet(z_refcat, t, ehs_nshot, is_target=False).sample     T:1374
            pred     = unet(z_tag, t, ehs, is_target=True).sample                T:1375
            loss = F.mse_loss(pred.float() + pred_ref.float() * 0., target)      T:1381-1384
            loss.backward(); clip_grad_norm_(unet.parameters(), 1.0); optimizer.step(); optimizer.zero_grad()
        The support call records its inputs and returns a graph-attached ZERO tensor of the right shape (the reference
        multiplies that output by 0); the query call runs the lock-step forward over [support ; query] and returns
        `.sample` with a grad_fn whose backward runs the tape with torch's d loss / d pred.  `parameters()[0].grad` is the
        flat fp32 gradient (a view of P.grad): None after optimizer.zero_grad() => the next step overwrites (first-touch
        writes), otherwise it accumulates (gradient accumulation, T:1323)."""
        from .unet import UNet2DConditionOutput
        for name, v in unused.items():
            if v is not None:
                raise NotImplementedError(f"{name} is not on the DiffewS hot path (always None there)")
        prm = self.parameters()[0]
        if not is_target:
            if sample.shape[1] != self.config["in_channels_ref"]:
                raise ValueError(f"support pass expects {self.config['in_channels_ref']} channels, got {sample.shape[1]}")
            self._pending_ref = (sample, timestep, encoder_hidden_states)
            out = _SupportPass.apply(prm, sample.shape[0], self.config["out_channels"], sample.shape[2], sample.shape[3])
        else:
            if sample.shape[1] != self.config["in_channels"]:
                raise ValueError(f"target pass expects {self.config['in_channels']} channels, got {sample.shape[1]}")
            ref = self._pending_ref
            self._pending_ref = None
            out = _QueryPass.apply(prm, self, ref, sample, timestep, encoder_hidden_states)
        return UNet2DConditionOutput(sample=out) if return_dict else (out,)

    def make_reducer(self, bucket_elems=54_000_000, comm_dtype=torch.float32, group=None, **kw):
        """GradBucketReducer over this trainer's flat gradient buffer (DDP's role, T:1226-1228): pass it to
        forward_backward(reducer=...) -- or set `self.reducer` for the autograd call surface -- and call .finish() before
        the optimizer step; .finish() returns the loss averaged over the ranks (T:1387)."""
        spec = {}
        for name, (off, shape) in self.P.spec.items():
            n = 1
            for d in shape:
                n *= d
            spec[name] = (off, n)
        return GradBucketReducer(self.P.grad_buf, spec, bucket_elems=bucket_elems, comm_dtype=comm_dtype, group=group, **kw)

    # ------------------------------------------------------------------ optimizer (T:1186-1194, T:1217-1223, T:1393-1394)
    def grad_sumsq(self):
        return ob.sumsq(self.P.grad)

    def _resolve_overflow(self):
        """Dynamic loss scale, GradScaler's rule (accelerate mixed_precision='fp16', T:1017): the previous optimizer step's
        overflow flag arrives through a pinned host word; on overflow that step was skipped on the device (dfw_adamw), so
        its step count is taken back and the scale halves; after `growth_interval` clean steps it doubles."""
        pend = self._overflow_pending
        if pend is None:
            return
        self._overflow_pending = None
        ev, host = pend
        ev.synchronize()
        if int(host[0]) != 0:
            self.step_count -= 1
            self.skipped_steps += 1
            self._good_steps = 0
            if self.dynamic_loss_scale:
                self.loss_scale = max(1.0, self.loss_scale * 0.5)
        else:
            self._good_steps += 1
            if self.dynamic_loss_scale and self._good_steps >= self.growth_interval:
                self.loss_scale = min(65536.0, self.loss_scale * 2.0)
                self._good_steps = 0

    def step_guard(self):
        """fp16 on the autograd call surface (`loss.backward()` + `torch.optim`, INTEGRATION.md path (a)): GradScaler's
        found-inf check, which accelerate performs inside `optimizer.step()` at T:1394 and which a plain torch.optim does
        NOT -- an overflowed fp16 step leaves inf / NaN in `parameters()[0].grad`, `clip_grad_norm_` turns the whole
        gradient into NaN and AdamW writes NaN into the fp32 master for good.  Call it between `loss.backward()` and
        `clip_grad_norm_`; it returns True when the gradient is finite (take the step).  On overflow it returns False --
        the caller skips `optimizer.step()` -- after zeroing the gradient and applying the dynamic loss scale's rule (halve
        now, double after `growth_interval` clean steps).  One host sync on a scalar.  `optimizer_step()` (path (b)) does
        the same on the device without the sync and needs no guard."""
        finite = bool(torch.isfinite(ob.sumsq(self.P.grad)).all())
        if finite:
            self._good_steps += 1
            if self.dynamic_loss_scale and self._good_steps >= self.growth_interval:
                self.loss_scale = min(65536.0, self.loss_scale * 2.0)
                self._good_steps = 0
        else:
            self.skipped_steps += 1
            self._good_steps = 0
            if self.dynamic_loss_scale:
                self.loss_scale = max(1.0, self.loss_scale * 0.5)
            self.P.grad.zero_()
        return finite

    def optimizer_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0):
        """clip_grad_norm_(max_grad_norm) + AdamW on the flat buffers; the 16-bit shadow is refreshed in the same pass.
        A non-finite gradient norm (fp16 overflow) SKIPS the update on the device -- master, moments and shadow are left
        untouched -- and is reported through `skipped_steps` / the dynamic loss scale one step later (no host sync here)."""
        P = self.P
        self._resolve_overflow()
        self.step_count += 1
        ss = ob.sumsq(P.grad)
        ob.adamw(P.master, P.grad, P.exp_avg, P.exp_avg_sq, self.step_count, lr, betas, eps, weight_decay, grad_sumsq=ss,
                 max_grad_norm=max_grad_norm or 0.0, shadow=P.shadow, found_inf=self._found_inf)
        host = self._found_host
        host.copy_(self._found_inf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._overflow_pending = (ev, host)
        P.version += 1
        return ss

    # ------------------------------------------------------------------ trainer state: resume (T:1281-1309, T:1407-1431)
    def optimizer_state_dict(self):
        """Everything `accelerator.save_state` keeps for the UNet besides the weights file: fp32 master, both AdamW
        moments (flat, packed layout), step count, loss-scale state.  `layout` pins the packed layout it belongs to."""
        self._resolve_overflow()
        P = self.P
        return {"layout": self._layout_id(), "step": self.step_count, "master": P.master.detach().cpu().clone(),
                "exp_avg": P.exp_avg.detach().cpu().clone(), "exp_avg_sq": P.exp_avg_sq.detach().cpu().clone(),
                "loss_scale": self.loss_scale, "good_steps": self._good_steps, "skipped_steps": self.skipped_steps}

    def load_optimizer_state_dict(self, sd):
        P = self.P
        if sd["layout"] != self._layout_id():
            raise ValueError("optimizer state belongs to a different parameter layout (model config / engine version)")
        P.master.copy_(sd["master"])
        P.exp_avg.copy_(sd["exp_avg"])
        P.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count, self.loss_scale = int(sd["step"]), float(sd["loss_scale"])
        self._good_steps, self.skipped_steps = int(sd.get("good_steps", 0)), int(sd.get("skipped_steps", 0))
        self._overflow_pending = None
        P.sync_shadow()
        if self._param is not None:
            self._master_seen = self._param._version

    def _layout_id(self):
        import hashlib
        h = hashlib.sha256()
        for name, (off, shape) in self.P.spec.items():
            h.update(f"{name}:{off}:{shape};".encode())
        return h.hexdigest()[:16]

    def save_state(self, path):
        """`accelerator.save_state(checkpoint-N)` (T:1407-1431): diffusers-format weights under unet/ (the hook of
        T:1130-1140) + the optimizer state, so that load_state() resumes the trajectory bit-for-bit."""
        import os
        os.makedirs(path, exist_ok=True)
        self.save_pretrained(path, subfolder="unet")
        torch.save(self.optimizer_state_dict(), os.path.join(path, "optimizer_dfw.pt"))

    def load_state(self, path):
        """`accelerator.load_state` (T:1281-1309)."""
        import os
        self.load_optimizer_state_dict(torch.load(os.path.join(path, "optimizer_dfw.pt"), map_location="cpu"))


class _SupportPass(torch.autograd.Function):
    """unet(z_refcat, t, ehs, is_target=False).sample: the reference multiplies it by zero (T:1381) -- a graph-attached
    zero tensor; its backward contributes nothing (the support rows' real gradient flows through the banks inside the
    query pass' backward)."""

    @staticmethod
    def forward(ctx, prm, n, oc, h, w):
        return prm.new_zeros(n, oc, h, w)

    @staticmethod
    def backward(ctx, g):
        return None, None, None, None, None


class _QueryPass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prm, trainer, ref, sample, timestep, ehs):
        fresh = prm.grad is None
        if ref is None:      # no support pass recorded: plain self-attention (0-shot)
            zr = sample.new_zeros(0, trainer.config["in_channels_ref"], *sample.shape[2:])
            ehs_ref = None
        else:
            zr, _, ehs_ref = ref
        c = trainer._forward(zr.detach(), sample.detach(), timestep, ehs.detach(), zero_grad=fresh,
                             ehs_ref=None if ehs_ref is None else ehs_ref.detach())
        ctx.trainer, ctx.c, ctx.prm = trainer, c, prm
        return c["pred"]

    @staticmethod
    def backward(ctx, g):
        tr, c = ctx.trainer, ctx.c
        n_ref = c["n_ref"]
        ob.loss_grad(g.to(torch.float32).contiguous(), tr.dtype, scale=tr.loss_scale, dpred_out=c["dpred"][n_ref:],
                     dpred_nchw_out=c["dpn"][n_ref:])
        if tr.reducer is not None:
            # the autograd call surface (`loss.backward()`, T:1390): the loss value lives in the caller's graph, not here, so
            # the tail slot carries 0 and the caller averages its own loss (T:1387 `accelerator.gather(loss)`); the buckets
            # fire during the tape walk exactly as under forward_backward(reducer=...)
            tr.reducer.begin(None)
        tr._backward(c, tr.reducer)
        ctx.prm.grad = tr.P.grad          # the flat fp32 gradient (packed layout), for clip_grad_norm_ / torch.optim
        ctx.c = None
        return None, None, None, None, None, None


def poly_lr(base_lr, step, total_steps, warmup_steps=0, lr_end=1e-7, power=1.0):
    """diffusers get_scheduler("polynomial", power=1.0) (T:1217-1223): linear warm-up, then polynomial decay."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    if step > total_steps:
        return lr_end
    remaining = 1 - (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return (base_lr - lr_end) * remaining ** power + lr_end


def _buckets(n, bucket_elems):
    return [(s0, min(n, s0 + bucket_elems)) for s0 in range(0, n, bucket_elems)]


def allreduce_flat_gradient(flat_grad, world_size=None, bucket_elems=54_000_000, group=None):
    """Gradient all-reduce of DDP (T:1226-1228, T:1391) on the flat fp32 gradient, SERIAL form: SUM over ranks in fixed
    buckets (216 MB each: sixteen cover the 866 M parameters), then the 1 / world_size average.  RCCL over xGMI on GPU
    tensors (backend "nccl"), gloo on CPU tensors.  Bucket boundaries depend only on the buffer length, so every rank
    issues the same sequence of collectives.  GradBucketReducer issues the same buckets DURING the backward."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    ws = world_size or dist.get_world_size(group)
    if ws == 1:
        return flat_grad
    for s0, s1 in _buckets(flat_grad.numel(), bucket_elems):
        dist.all_reduce(flat_grad[s0:s1], op=dist.ReduceOp.SUM, group=group)
        flat_grad[s0:s1].mul_(1.0 / ws)
    return flat_grad


class _SegmentRecorder:
    """Stands in for a GradBucketReducer while UNetTrainer.forward_backward is CAPTURED: same begin / mark / finish calls, but
    instead of issuing a bucket's collective when its last writer is in the stream it ENDS the HIP graph being captured
    there and begins the next one.  Result: `segments` = [(graph, [bucket, ...]), ...] -- replaying graph i and then firing
    its buckets reproduces the eager overlapped step launch for launch (replay_segments)."""

    def __init__(self, reducer, pool):
        self.red, self.pool = reducer, pool
        self.segments, self.cur = [], None

    def _open(self):
        self.cur = torch.cuda.CUDAGraph(keep_graph=True)
        self.cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def _close(self, buckets):
        self.cur.capture_end()
        self.segments.append((self.cur, list(buckets)))
        self.cur = None

    def begin(self, loss=None):
        r = self.red
        r.reset_state()
        slot = r.buf[r.loss_slot:r.loss_slot + 1]
        if loss is None:
            raise ValueError("the segmented step carries the loss in the last bucket: begin(loss)")
        slot.copy_(loss.reshape(1).to(r.buf.dtype))          # captured in the first segment (a copy kernel, not a memset)

    def would_fire(self, names):
        return self.red.would_fire(names)

    def mark(self, names):
        ready = self.red.ready_after(names)
        if ready:
            for b in ready:
                self.red.fired[b] = True
            self._close(ready)
            self._open()

    def finish(self):
        rest = [b for b in range(len(self.red.ranges) - 1, -1, -1) if not self.red.fired[b]]
        for b in rest:
            self.red.fired[b] = True
        self._close(rest)
        self.red.active = False


def replay_segments(segments, reducer):
    """One training micro-step as a chain of captured segments: replay segment i (its kernels enter the compute stream), then
    issue the all-reduce of the gradient buckets whose last writer was in it -- from the reducer's comm stream, behind an
    event, while the compute stream goes on with segment i + 1.  Same buckets, same order, same arithmetic as the eager
    overlapped step and as the serial allreduce_flat_gradient (bit-identical in fp32).  Returns reducer.finish()'s loss."""
    reducer.reset_state()
    for graph, buckets in segments:
        graph.replay()
        for b in buckets:
            if not reducer.fired[b]:
                reducer._fire(b)
    return reducer.finish()


class GradBucketReducer:
    """DDP's overlapped gradient all-reduce (T:1226-1228, T:1391) for the flat gradient buffer.

    The flat layout follows the forward's order of use (UNetTrainer._build), so the backward finishes parameter
    gradients from the END of the buffer towards its start and the fixed buckets of allreduce_flat_gradient complete one
    after the other.  mark(names) is called after the launches that wrote those parameters' gradients were issued; when
    the last writer of a bucket is in the compute stream, an event is recorded there, the COMM stream waits for it and
    issues the bucket's all-reduce (+ the 1 / world average) -- while the compute stream goes on with the backward of the
    earlier layers.  finish() issues whatever is left, makes the compute stream wait for the comm stream and returns the
    rank-averaged loss: the scalar rides in the tail slot of the LAST bucket range (one collective fewer than T:1387's
    gather).  Same buckets, same reduction operator and the same elementwise average as the serial form => bit-identical
    results in fp32.  comm_dtype=torch.bfloat16 halves the bytes on xGMI: bucket -> bf16 (dfw_convert_f32), SUM in bf16
    on the wire, back to fp32 times 1 / world (dfw_convert_to_f32).

    spec: {name: (offset, numel)} of the parameters inside `buf`; buf = P.grad_buf (numel + TAIL floats).
    collective: injectable for tests -- f(tensor) reduces in place."""

    def __init__(self, buf, spec, bucket_elems=54_000_000, comm_dtype=torch.float32, group=None, world_size=None,
                 collective=None, loss_slot=None):
        import torch.distributed as dist
        self.buf, self.group = buf, group
        self.comm_dtype = comm_dtype
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.ws = world_size or (dist.get_world_size(group) if self.dist else 1)
        self.collective = collective
        self.ranges = _buckets(buf.numel(), bucket_elems)
        self.loss_slot = buf.numel() - ParamStore.TAIL if loss_slot is None else loss_slot
        self.of = {}                                  # name -> bucket indices it overlaps
        self.need = [set() for _ in self.ranges]
        for name, (off, n) in spec.items():
            b0, b1 = off // bucket_elems, (off + max(n, 1) - 1) // bucket_elems
            self.of[name] = list(range(b0, b1 + 1))
            for b in self.of[name]:
                self.need[b].add(name)
        self.cuda = buf.is_cuda
        self.comm = torch.cuda.Stream(device=buf.device) if self.cuda else None
        self.fired_order = []
        self.active = False
        self.steps_begun = 0                          # begin() calls so far (tests / launch-loop sanity checks)

    def reset_state(self):
        """Host-side bookkeeping of begin() alone (no device work): the segmented graph replay re-arms the reducer with it,
        the loss slot having been written by the first captured segment."""
        self.left = [set(s) for s in self.need]
        self.fired = [False] * len(self.ranges)
        self.fired_order = []
        self.active = True
        self.steps_begun += 1

    def would_fire(self, names):
        """True if reporting `names` now would complete at least one bucket (nothing is changed)."""
        ns = set(names)
        for name in names:
            for b in self.of.get(name, ()):
                if not self.fired[b] and self.left[b] <= ns:
                    return True
        return False

    def ready_after(self, names):
        """Buckets (indices, in firing order) that become complete once `names` are final -- mark() without the firing;
        the segment recorder cuts the captured step there."""
        out = []
        for name in names:
            for b in self.of.get(name, ()):
                self.left[b].discard(name)
                if not self.left[b] and not self.fired[b] and b not in out:
                    out.append(b)
        return out

    def begin(self, loss=None):
        """Start of a backward walk.  loss: the scalar to average with the last bucket (None: the slot carries 0)."""
        self.reset_state()
        slot = self.buf[self.loss_slot:self.loss_slot + 1]
        if loss is not None:                          # before any bucket can fire: the loss is known ahead of the backward
            slot.copy_(loss.reshape(1).to(self.buf.dtype))
        else:
            slot.zero_()

    def mark(self, names):
        if not self.active:
            # a reducer that is attached to a backward walk but was never begun would let every rank keep its LOCAL
            # gradient without any error (the ranks then diverge silently): refuse
            raise RuntimeError("GradBucketReducer.mark() without begin(): the gradient buckets of this step would never be "
                               "reduced -- call begin() before the backward (forward_backward(reducer=...) and the autograd "
                               "surface with `trainer.reducer` set do)")
        for name in names:
            for b in self.of.get(name, ()):
                self.left[b].discard(name)
                if not self.left[b] and not self.fired[b]:
                    self._fire(b)

    def _reduce(self, t):
        if self.collective is not None:
            self.collective(t)
        elif self.dist is not None and self.ws > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _fire(self, b):
        self.fired[b] = True
        self.fired_order.append(b)
        s0, s1 = self.ranges[b]
        sl = self.buf[s0:s1]
        if self.ws == 1 and self.collective is None:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.buf.device))      # after the bucket's last writer
            self.comm.wait_event(ev)
            ctx = torch.cuda.stream(self.comm)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if self.comm_dtype == torch.float32:
                self._reduce(sl)
                sl.mul_(1.0 / self.ws)
            else:
                n8 = (sl.numel() + 7) // 8 * 8
                if self.cuda and sl.numel() == n8 and sl.data_ptr() % 16 == 0:
                    wire = ops.to_storage(sl, self.comm_dtype)
                    self._reduce(wire)
                    ob.to_f32(wire, sl, scale=1.0 / self.ws)
                else:                                  # CPU tensors (gloo tests) / ragged tail bucket
                    wire = sl.to(self.comm_dtype)
                    self._reduce(wire)
                    sl.copy_(wire.to(torch.float32) * (1.0 / self.ws))
                if self.cuda:
                    wire.record_stream(self.comm)

    def finish(self):
        """Issue the buckets that are still open (parameters nobody wrote this step), then join the streams.
        Returns the rank-averaged loss tensor [1] (a view of the tail slot)."""
        if not self.active:
            raise RuntimeError("GradBucketReducer.finish() without a begun step: no backward walk reported to this reducer "
                               "since the last finish() -- the gradient in the buffer has NOT been reduced")
        for b in range(len(self.ranges) - 1, -1, -1):
            if not self.fired[b]:
                self._fire(b)
        self.active = False
        if self.cuda:
            torch.cuda.current_stream(self.buf.device).wait_stream(self.comm)
        return self.buf[self.loss_slot:self.loss_slot + 1]
