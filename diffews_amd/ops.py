"""Host-side op wrappers: torch tensors (device memory + current stream only) -> C-ABI calls.

Every function launches hand-written gfx950 kernels from libdiffews_hip.so on
`torch.cuda.current_stream()`; nothing here computes with PyTorch.  Activations are NHWC
(`[B, H, W, C]` or `[rows, C]`) in the engine storage dtype (bf16 or fp16).
"""
import ctypes as C
import os
import math

import torch

from . import _lib as L

_DT = {torch.bfloat16: L.BF16, torch.float16: L.F16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"engine storage dtype must be bfloat16 or float16, got {t.dtype}")


def _f32(t, name):
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous float32 tensor")
    return t


def _rowbias(a, rb):
    """rowbias [imgs, N] fp32, rows may be strided (a column slice of a wider buffer)."""
    if rb is None:
        return
    if rb.dtype != torch.float32 or rb.dim() != 2 or rb.stride(1) != 1:
        raise TypeError("rowbias must be float32 [imgs, N] with unit column stride")
    a.rowbias, a.ld_rowbias = rb.data_ptr(), rb.stride(0)


# Optional per-launch timing hook (bench.py roofline leg): when set, every GEMM launch is bracketed
# by events on the launch stream and reported as hook(kernel_name, flops, start_event, end_event).
gemm_hook = None


def _gemm_call(a):
    lib = L.lib()
    if gemm_hook is not None:
        buf = C.create_string_buffer(64)
        L.check(lib.dfw_gemm_kernel_name(C.byref(a), buf, 64), "dfw_gemm_kernel_name")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _gemm_launch(lib, a)
        e1.record()
        gemm_hook(buf.value.decode(), 2.0 * a.M * a.N * a.K * max(1, a.batch), e0, e1,
                  (a.M, a.N, a.K, a.taps, a.stride, a.ups, a.splitk, max(1, a.batch)))
        return
    _gemm_launch(lib, a)


def _gemm_launch(lib, a):
    nbytes = lib.dfw_gemm_workspace_bytes(C.byref(a))   # 0 unless the plan uses split-K
    if nbytes:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    L.check(lib.dfw_gemm(C.byref(a), _stream()), "dfw_gemm")
    return


def linear(x, w, bias=None, residual=None, rowbias=None, rows_per_img=0, act=L.ACT_NONE, geglu=False,
           out=None, out_f32=False, out_scale=1.0, splitk=None, colscale=None):
    """y[M, N] = epi(x[M, K] @ w[N, K]^T).  x may be a row-strided view ([M, K] with stride (ld, 1)).
    colscale = (n, s): output columns < n (a multiple of 64) are multiplied by s in fp32 before rounding."""
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.is_contiguous()
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    assert out.stride(1) == 1 and out.shape == (M, n_out)
    a = L.GemmArgs()
    a.A, a.W, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    a.bias = _p(_f32(bias, "bias"))
    _rowbias(a, rowbias)
    if residual is not None:
        assert residual.dtype in (x.dtype, torch.float32) and residual.stride(1) == 1 and residual.shape == (M, N)
        a.residual, a.ldr = residual.data_ptr(), residual.stride(0)
        a.residual_f32 = int(residual.dtype == torch.float32)   # the fp32 residual stream
    a.a_elems = (M - 1) * x.stride(0) + K
    a.w_elems = w.numel()
    a.M, a.N, a.K, a.lda, a.ldc = M, N, K, x.stride(0), out.stride(0)
    a.taps, a.Cin = 1, K
    a.rows_per_img = rows_per_img
    a.out_scale, a.act, a.geglu = out_scale, act, int(geglu)
    a.out_mode = L.OUT_F32 if out.dtype == torch.float32 else L.OUT_T
    a.splitk = 0 if splitk is None else splitk   # 0: the library plans tile + split-K
    a.batch, a.dtype = 1, _dt(x)
    if colscale is not None:
        a.colscale_n, a.colscale = int(colscale[0]), float(colscale[1])
    _gemm_call(a)
    return out


def bmm_nt(x, w, out_f32=False, out_scale=1.0):
    """Batched y[b] = x[b] @ w[b]^T for contiguous x [Bt, M, K], w [Bt, N, K]."""
    assert x.dim() == 3 and w.dim() == 3 and x.is_contiguous() and w.is_contiguous()
    Bt, M, K = x.shape
    N = w.shape[1]
    out = torch.empty(Bt, M, N, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    a = L.GemmArgs()
    a.A, a.W, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    a.a_elems, a.w_elems = M * K, N * K
    a.M, a.N, a.K, a.lda, a.ldc = M, N, K, K, N
    a.taps, a.Cin = 1, K
    a.out_scale = out_scale
    a.out_mode = L.OUT_F32 if out_f32 else L.OUT_T
    a.splitk, a.batch = 1, Bt
    a.strideA, a.strideW, a.strideC = M * K, N * K, M * N
    a.dtype = _dt(x)
    _gemm_call(a)
    return out


def conv3x3(x, w, cout, bias=None, stride=1, pad=1, ups=False, rowbias=None, residual=None,
            out_nchw_f32=False, out_scale=1.0, splitk=None, gn_groups=0, gn_in=None, act=L.ACT_NONE,
            out_f32=False, w_blk=None):
    """3x3 conv on NHWC x [B, H, W, Cin] with w packed [Cout, 9*Cin] (ky, kx, cin order).
    w_blk: optional blocked copy of w (packing.block_conv3x3: [9][Cin/32][Cout][32]) for the kernels that stage W per
    (tap, 32-channel chunk): their LDS-DMA then fetches whole cache lines (dfw_gemm_args.W_blocked).
    pad = top/left zero padding (bottom/right come from bounds checks: pad=0,stride=2 is the VAE
    encoder's F.pad(0,1,0,1) + conv(stride 2, padding 0)); ups fuses nearest-2x upsampling.
    gn_in = (gamma, beta, groups, eps, silu): the conv's input is GroupNorm(+SiLU) of x, applied by dfw_groupnorm first
    (ResnetBlock2D's norm + nonlinearity + conv as one call; a kernel that normalised its input patch in LDS instead was
    built in rounds 1-3, never beat pass + conv, and is gone: DESIGN.md section 3).
    out_f32: NHWC fp32 output, and `residual` may be fp32 -- the fp32 residual stream (x + branch summed and stored
    in fp32; the conv's operands stay 16-bit)."""
    assert x.dim() == 4 and x.stride(3) == 1 and x.is_contiguous()
    B, Hi, Wi, Cin = x.shape
    if gn_in is not None:
        x = groupnorm(x, *gn_in, out_dtype=w.dtype)
    assert w.shape == (cout, 9 * Cin) and w.dtype == x.dtype and w.is_contiguous()
    if ups:
        assert stride == 1 and pad == 1
        Ho, Wo = 2 * Hi, 2 * Wi
    elif stride == 1:
        Ho, Wo = Hi, Wi
    else:
        Ho, Wo = (Hi + 2 * pad - 3) // stride + 1 if pad else Hi // 2, (Wi + 2 * pad - 3) // stride + 1 if pad else Wi // 2
    M = B * Ho * Wo
    if out_nchw_f32:
        out = torch.empty(B, cout, Ho, Wo, dtype=torch.float32, device=x.device)
    else:
        out = torch.empty(B, Ho, Wo, cout, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    a = L.GemmArgs()
    a.A, a.W, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    if w_blk is not None:
        assert w_blk.dtype == w.dtype and w_blk.numel() == w.numel() and w_blk.is_contiguous() and Cin % 32 == 0
        a.W_blocked = w_blk.data_ptr()
    a.bias = _p(_f32(bias, "bias"))
    _rowbias(a, rowbias)
    if residual is not None:
        assert residual.dtype in (x.dtype, torch.float32) and residual.is_contiguous() and residual.numel() == M * cout
        a.residual, a.ldr = residual.data_ptr(), cout
        a.residual_f32 = int(residual.dtype == torch.float32)
    a.a_elems, a.w_elems = x.numel(), w.numel()
    a.M, a.N, a.K, a.lda, a.ldc = M, cout, 9 * Cin, Cin, cout
    a.taps, a.Cin, a.Hi, a.Wi, a.Ho, a.Wo = 9, Cin, Hi, Wi, Ho, Wo
    a.stride, a.pad, a.ups = stride, pad, int(ups)
    a.rows_per_img = Ho * Wo
    a.out_scale, a.act = out_scale, act
    a.out_mode = L.OUT_NCHW_F32 if out_nchw_f32 else (L.OUT_F32 if out_f32 else L.OUT_T)
    a.splitk = 0 if splitk is None else splitk
    a.batch, a.dtype = 1, _dt(x)
    stats = None
    if gn_groups and not out_nchw_f32:
        # fused GroupNorm statistics of the output, when the planned kernel supports them
        a.gn_groups = gn_groups
        chunks = L.lib().dfw_gemm_gn_chunks(C.byref(a))
        if chunks > 0:
            part = torch.empty(B, chunks, gn_groups, 2, dtype=torch.float32, device=x.device)
            a.gn_partial = part.data_ptr()
            stats = (part, chunks, gn_groups)
    _gemm_call(a)
    if stats is not None:
        out._gn_stats = stats   # consumed by groupnorm(out, ...) -- valid while `out` is not modified
    return out




def fsa_attention(q, k, v, heads, k_bank=None, v_bank=None, nshot=0, scale=None, out=None, n_plain=0,
                  q_prescaled=False, lse=None, key_split=True):
    """KV-fusion self-attention.  q/k/v: [B, N, heads*64] views (token stride = stride(1));
    k_bank/v_bank: [(B-n_plain)*nshot, Nb, heads*64] views written by the support pass.
    n_plain: the first n_plain batch entries ignore the bank (lock-step [support ; query] launch).
    q_prescaled: q already carries scale * log2(e) (linear(..., colscale=(C, FSA_QSCALE))).
    key_split: let the library split the bank readers' key range over several workgroups when that balances the launch
    (many shots; needs a scratch buffer, allocated here)."""
    B, N, Cq = q.shape
    assert Cq == heads * 64 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty(B, N, Cq, dtype=q.dtype, device=q.device)
    assert out.shape == (B, N, Cq) and out.stride(2) == 1 and out.stride(1) == Cq
    a = L.FsaArgs()
    a.q, a.k, a.v, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.batch, a.heads, a.n_q, a.n_kv = B, heads, N, k.shape[1]
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), Cq
    a.q_bs, a.k_bs, a.v_bs, a.o_bs = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    if nshot:
        assert k_bank.shape[0] == (B - n_plain) * nshot and k_bank.stride(2) == 1 and v_bank.stride(2) == 1
        assert k_bank.dtype == q.dtype and v_bank.shape == k_bank.shape
        a.k_bank, a.v_bank = k_bank.data_ptr(), v_bank.data_ptr()
        a.n_bank, a.nshot = k_bank.shape[1], nshot
        a.ldkb, a.ldvb, a.kb_bs, a.vb_bs = k_bank.stride(1), v_bank.stride(1), k_bank.stride(0), v_bank.stride(0)
    a.scale = scale if scale is not None else 64 ** -0.5
    a.dtype, a.n_plain, a.q_prescaled = _dt(q), n_plain, int(bool(q_prescaled))
    if lse is not None:   # training: per-row log2-sum-exp2 for the backward
        assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.shape == (B, heads, N)
        a.lse = lse.data_ptr()
    nbytes = L.lib().dfw_fsa_workspace_bytes(C.byref(a)) if (nshot >= 2 and key_split) else 0   # key split of the bank readers (many shots)
    if nbytes:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    if gemm_hook is not None:   # bench.py roofline leg: QK^T + PV flops of this launch
        keys = n_plain * k.shape[1] + (B - n_plain) * (k.shape[1] + (nshot * k_bank.shape[1] if nshot else 0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().dfw_fsa_attention(C.byref(a), _stream()), "dfw_fsa_attention")
        e1.record()
        gemm_hook("fsa_attention", 4.0 * heads * 64 * N * keys, e0, e1, (B, heads, N, keys))
        return out
    L.check(L.lib().dfw_fsa_attention(C.byref(a), _stream()), "dfw_fsa_attention")
    return out


def zeros(shape, dtype, device="cuda"):
    """torch.zeros whose fill is a library kernel (capture-safe: no memset node).  Sizes are padded to 16 bytes."""
    n = 1
    for d in shape:
        n *= int(d)
    esz = torch.empty(0, dtype=dtype).element_size()
    pad = (-(n * esz)) % 16 // esz if (n * esz) % 16 else 0
    buf = torch.empty(n + pad, dtype=dtype, device=device)
    L.check(L.lib().dfw_zero(buf.data_ptr(), (n + pad) * esz, _stream()), "dfw_zero")
    return buf[:n].view(*shape)


def split_storage(x, dtype):
    """fp32 x -> (hi, lo) in `dtype` with hi + lo = x to ~2^-22 relative (fp16): see linear_stream / conv3x3_stream."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 8 == 0
    hi = torch.empty(x.shape, dtype=dtype, device=x.device)
    lo = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().dfw_split_f32(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), _DT[dtype], _stream()), "dfw_split_f32")
    return hi, lo


def linear_stream(x, w, bias=None, residual=None):
    """Linear whose INPUT is the residual stream (ResnetBlock2D.conv_shortcut).  16-bit stream: plain linear.  fp32 stream:
    x is fed as two 16-bit operands (hi, lo = split_storage(x)); the second GEMM adds onto the first through the fp32 residual
    epilogue, so the stream's 16-bit rounding never enters the fp32 result."""
    if x.dtype != torch.float32:
        return linear(x, w, bias=bias, residual=residual)
    hi, lo = split_storage(x, w.dtype)
    y = linear(hi, w, bias=bias, residual=residual, out_f32=True)
    return linear(lo, w, residual=y, out_f32=True)


def conv3x3_stream(x, w, cout, bias=None, lo=True, **kw):
    """conv3x3 whose INPUT is the residual stream (Downsample2D / Upsample2D convs): as linear_stream.
    lo=False: the fp32 stream is fed as ONE operand rounded once (no second GEMM, a convert instead of the split): each such
    conv adds one storage rounding of its input (2.5e-4 relative in fp16, in quadrature) -- the VAE encoder's two LARGEST
    downsample convs take this form (round 4): they were 2.2 ms of the parity mode for 0.8e-4 of its 7.5e-4."""
    if x.dtype != torch.float32:
        return conv3x3(x, w, cout, bias=bias, **kw)
    if not lo:
        return conv3x3(to_storage(x, w.dtype), w, cout, bias=bias, out_f32=True, **kw)
    hi, lo = split_storage(x, w.dtype)
    gn_groups = kw.pop("gn_groups", 0)
    y = conv3x3(hi, w, cout, bias=bias, out_f32=True, **kw)
    return conv3x3(lo, w, cout, residual=y, out_f32=True, gn_groups=gn_groups, **kw)     # the FINAL values carry the statistics


FSA_QSCALE = 64 ** -0.5 * math.log2(math.e)   # attn.scale (A:269-271, head_dim 64) in exp2 units


VATTN_QSCALE = 512 ** -0.5 * math.log2(math.e)   # the VAE mid-block attention's scale (one head of dim 512) in exp2 units


def vae_attention(q, k, v, out=None):
    """Flash attention of the VAE mid-block: q / k / v [B, N, 512] views (column slices of one fused QKV buffer), ONE head of
    dim 512, q pre-multiplied by VATTN_QSCALE (linear(..., colscale=(512, VATTN_QSCALE))).  The N x N scores stay on chip."""
    B, N, D = q.shape
    assert D == 512 and k.shape == q.shape and v.shape == q.shape and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty(B, N, D, dtype=q.dtype, device=q.device)
    a = L.VattnArgs()
    a.q, a.k, a.v, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.batch, a.n, a.head_dim, a.q_prescaled = B, N, D, 1
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    a.q_bs, a.k_bs, a.v_bs, a.o_bs = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.dtype = _dt(q)
    L.check(L.lib().dfw_vae_attention(C.byref(a), _stream()), "dfw_vae_attention")
    return out


def cross_attention(q, k, v, heads, scale=None):
    """q [B, N, heads*64]; k/v [B, L, heads*64] views (short context)."""
    B, N, Cq = q.shape
    assert Cq == heads * 64 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    out = torch.empty(B, N, Cq, dtype=q.dtype, device=q.device)
    a = L.XattnArgs()
    a.q, a.k, a.v, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.batch, a.heads, a.n_q, a.L = B, heads, N, k.shape[1]
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), Cq
    a.q_bs, a.k_bs, a.v_bs, a.o_bs = q.stride(0), k.stride(0), v.stride(0), N * Cq
    a.scale = scale if scale is not None else 64 ** -0.5
    a.dtype = _dt(q)
    L.check(L.lib().dfw_cross_attention(C.byref(a), _stream()), "dfw_cross_attention")
    return out


def groupnorm(x, gamma, beta, groups, eps, silu=False, return_stats=False, out_dtype=None):
    """GroupNorm (+SiLU) over NHWC x [B, H, W, C] (or [B, HW, C]).
    return_stats: also return the (mean, rstd) [B, groups, 2] fp32 the kernel normalised with (training).
    x may be fp32 (the fp32 residual stream): the output is then `out_dtype` (bf16 / fp16, required)."""
    assert x.is_contiguous()
    xf32 = x.dtype == torch.float32
    if xf32 and out_dtype not in _DT:
        raise TypeError("groupnorm of an fp32 tensor needs out_dtype=torch.bfloat16 / torch.float16")
    odt = out_dtype if xf32 else x.dtype
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    a = L.GroupNormArgs()
    y = torch.empty(x.shape, dtype=odt, device=x.device)
    a.x, a.y, a.gamma, a.beta = x.data_ptr(), y.data_ptr(), _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta"))
    a.B, a.HW, a.C, a.groups, a.ldx, a.ldy = B, HW, Cc, groups, Cc, Cc
    a.eps, a.silu, a.dtype, a.x_f32 = eps, int(silu), _DT[odt], int(xf32)
    st = getattr(x, "_gn_stats", None)
    if st is not None and st[2] == groups and st[0].shape[0] == B:
        a.pre_partial, a.pre_chunks = st[0].data_ptr(), st[1]
    lib = L.lib()
    nbytes = lib.dfw_groupnorm_workspace_bytes(C.byref(a))
    if nbytes == 0:
        L.check(L.lib().dfw_groupnorm(C.byref(a), _stream()), "dfw_groupnorm")  # reports the shape error
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    a.stats_ws, a.stats_ws_bytes = ws.data_ptr(), nbytes
    L.check(lib.dfw_groupnorm(C.byref(a), _stream()), "dfw_groupnorm")
    if return_stats:
        return y, ws[ws.numel() - B * groups * 2:].view(B, groups, 2)
    return y


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=None):
    """x may be fp32 (the fp32 residual stream inside a transformer block): the output is then `out_dtype`."""
    assert x.dim() == 2 and x.stride(1) == 1
    xf32 = x.dtype == torch.float32
    if xf32 and out_dtype not in _DT:
        raise TypeError("layernorm of an fp32 tensor needs out_dtype=torch.bfloat16 / torch.float16")
    odt = out_dtype if xf32 else x.dtype
    y = torch.empty(x.shape, dtype=odt, device=x.device)
    a = L.LayerNormArgs()
    a.x, a.y, a.gamma, a.beta = x.data_ptr(), y.data_ptr(), _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta"))
    a.rows, a.C, a.ldx, a.ldy, a.eps, a.dtype = x.shape[0], x.shape[1], x.stride(0), y.stride(0), eps, _DT[odt]
    a.x_f32 = int(xf32)
    L.check(L.lib().dfw_layernorm(C.byref(a), _stream()), "dfw_layernorm")
    return y


def conv_small(x, w, bias, cout, taps, dtype, nchw_f32_out=False, in_scale=1.0, out_scale=1.0, gn_groups=0,
               out=None, gn_part=None, out_f32=False):
    """Boundary conv with Cin <= 8: x NCHW fp32 [B, Cin, H, W], w fp32 [Cout, taps, Cin].
    gn_groups: also emit the GroupNorm partial sums of the output where the kernel supports it (y._gn_stats).
    x may be a list/tuple of up to three such tensors (same Cin, H, W): the batch is their concatenation,
    read in place (no torch.cat).
    out: write into this tensor instead of allocating -- NHWC: a batch slice of a larger contiguous buffer;
    NCHW fp32: a view [B, cout, H, W] whose channel planes are contiguous (a channel slice of a wider NCHW
    tensor is fine: the batch stride travels as y_bstride).  gn_part: the matching [B, chunks, groups, 2]
    slice of a larger partial-sum buffer (several conv_small calls filling one batch)."""
    xs = list(x) if isinstance(x, (list, tuple)) else [x]
    assert 1 <= len(xs) <= 3
    for t in xs:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4 and t.shape[1:] == xs[0].shape[1:]
    x = xs[0]
    _, Cin, H, W = x.shape
    B = sum(t.shape[0] for t in xs)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == cout * taps * Cin
    a = L.ConvSmallArgs()
    if len(xs) > 1:
        a.x1, a.b0 = xs[1].data_ptr(), xs[0].shape[0]
        a.b1 = a.b0 + xs[1].shape[0]
        if len(xs) > 2:
            a.x2 = xs[2].data_ptr()
    if nchw_f32_out:
        y = out if out is not None else torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
        assert y.shape == (B, cout, H, W) and y.dtype == torch.float32
        assert y.stride(3) == 1 and y.stride(2) == W and y.stride(1) == H * W
        a.y_bstride = y.stride(0) if B > 1 else cout * H * W
    else:   # NHWC: storage dtype, or fp32 (out_f32: the fp32 residual stream starts at conv_in)
        ydt = torch.float32 if out_f32 else dtype
        y = out if out is not None else torch.empty(B, H, W, cout, dtype=ydt, device=x.device)
        assert y.shape == (B, H, W, cout) and y.dtype == ydt and y.is_contiguous()
    a.x, a.W, a.bias, a.y = x.data_ptr(), w.data_ptr(), _p(_f32(bias, "bias")), y.data_ptr()
    a.B, a.Cin, a.H, a.Wd, a.Cout, a.taps, a.ldy = B, Cin, H, W, cout, taps, cout
    a.in_scale, a.out_scale = in_scale, out_scale
    a.out_mode = L.OUT_NCHW_F32 if nchw_f32_out else (L.OUT_F32 if out_f32 else L.OUT_T)
    a.dtype = _DT[dtype]
    stats = None
    if gn_groups and not nchw_f32_out and not out_f32:
        a.gn_groups = gn_groups
        chunks = L.lib().dfw_conv_small_gn_chunks(C.byref(a))
        if chunks > 0:
            part = gn_part if gn_part is not None else torch.empty(B, chunks, gn_groups, 2, dtype=torch.float32, device=x.device)
            assert part.shape == (B, chunks, gn_groups, 2) and part.is_contiguous() and part.dtype == torch.float32
            a.gn_partial = part.data_ptr()
            stats = (part, chunks, gn_groups)
    L.check(L.lib().dfw_conv_small(C.byref(a), _stream()), "dfw_conv_small")
    if stats is not None:
        y._gn_stats = stats
    return y


def conv_small_gn_chunks(x_shape, cout, taps, dtype, gn_groups):
    """Chunks per image of conv_small's fused GroupNorm partial sums for this shape (0: unsupported)."""
    B, Cin, H, W = x_shape
    a = L.ConvSmallArgs()
    a.B, a.Cin, a.H, a.Wd, a.Cout, a.taps, a.ldy = B, Cin, H, W, cout, taps, cout
    a.out_mode, a.dtype, a.gn_groups = L.OUT_T, _DT[dtype], gn_groups
    a.x = 16    # alignment of the input pointer is part of the kernel choice: torch allocations are 16-byte aligned
    return L.lib().dfw_conv_small_gn_chunks(C.byref(a))


def meter_update(counts, class_id, inter_buf, union_buf):
    """AverageMeter.update on device: int64 atomics into the two [2, nclass] buffers (logger.py:35-37)."""
    assert counts.dtype == torch.int64 and counts.is_contiguous() and counts.shape[1] == 4
    assert class_id.dtype == torch.int64 and class_id.is_contiguous() and class_id.shape[0] == counts.shape[0]
    assert inter_buf.dtype == torch.int64 and inter_buf.is_contiguous() and union_buf.is_contiguous()
    L.check(L.lib().dfw_meter_update(counts.data_ptr(), class_id.data_ptr(), inter_buf.data_ptr(), union_buf.data_ptr(),
                                     counts.shape[0], inter_buf.shape[1], _stream()), "dfw_meter_update")


def softmax_rows(x, dtype, scale=1.0):
    assert x.dtype == torch.float32 and x.is_contiguous()
    Lr = x.shape[-1]
    rows = x.numel() // Lr
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().dfw_softmax_rows(x.data_ptr(), y.data_ptr(), rows, Lr, scale, _DT[dtype], _stream()),
            "dfw_softmax_rows")
    return y


def softmax_groups(x, groups, L_, dtype):
    """x [rows, ld] fp32 scores -> [rows, ld] `dtype`: softmax over each group's L entries, padding zeroed."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().dfw_softmax_groups(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], groups, L_,
                                       _DT[dtype], _stream()), "dfw_softmax_groups")
    return y


def transpose(x):
    assert x.dim() == 3 and x.is_contiguous()
    Bt, R, Cc = x.shape
    y = torch.empty(Bt, Cc, R, dtype=x.dtype, device=x.device)
    L.check(L.lib().dfw_transpose(x.data_ptr(), y.data_ptr(), Bt, R, Cc, _dt(x), _stream()), "dfw_transpose")
    return y


def concat_channels(a, b):
    assert a.is_contiguous() and b.is_contiguous() and a.shape[:-1] == b.shape[:-1] and a.dtype == b.dtype
    Ca, Cb = a.shape[-1], b.shape[-1]
    y = torch.empty(*a.shape[:-1], Ca + Cb, dtype=a.dtype, device=a.device)
    rows = a.numel() // Ca
    if a.dtype == torch.float32:   # fp32 residual stream: a byte copy -- C fp32 channels are 2C 16-bit units to the kernel
        L.check(L.lib().dfw_concat_channels(a.data_ptr(), b.data_ptr(), y.data_ptr(), rows, 2 * Ca, 2 * Cb, L.BF16, _stream()),
                "dfw_concat_channels")
        return y
    L.check(L.lib().dfw_concat_channels(a.data_ptr(), b.data_ptr(), y.data_ptr(), rows, Ca, Cb, _dt(a), _stream()),
            "dfw_concat_channels")
    return y


def to_storage(x, dtype):
    """fp32 -> storage dtype copy (the 16-bit MFMA-operand view of an fp32 residual-stream tensor); identity for a
    tensor that already is `dtype`."""
    if x.dtype == dtype:
        return x
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 8 == 0
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().dfw_convert_f32(x.data_ptr(), y.data_ptr(), x.numel(), _DT[dtype], _stream()), "dfw_convert_f32")
    return y


def timestep_embedding(timesteps, dim, dtype, flip_sin_to_cos=True, freq_shift=0.0):
    assert timesteps.dtype == torch.float32 and timesteps.is_contiguous() and timesteps.dim() == 1
    out = torch.empty(timesteps.shape[0], dim, dtype=dtype, device=timesteps.device)
    L.check(L.lib().dfw_timestep_embedding(timesteps.data_ptr(), out.data_ptr(), timesteps.shape[0], dim,
                                           int(flip_sin_to_cos), float(freq_shift), _DT[dtype], _stream()),
            "dfw_timestep_embedding")
    return out


def seg_postprocess(x, gt=None, r_threshold=0.25, threshold=0.0, batch_max=False, u8_out=None, counts_out=None,
                    scratch=None):
    """x: decoder output [B, 3, H, W] fp32 -> (uint8 [B,3,H,W], counts int64 [B,4] or None).
    r_threshold > 0: dynamic threshold r_threshold * max (per image, or over the batch tensor when
    batch_max -- main_oss.py:131 read literally); else the fixed `threshold` (main_oss.py:134-135)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[1] == 3
    B, _, H, W = x.shape
    u8 = u8_out if u8_out is not None else torch.empty(B, 3, H, W, dtype=torch.uint8, device=x.device)
    if scratch is None:
        scratch = torch.empty(B, dtype=torch.int32, device=x.device)
    counts = None
    if gt is not None:
        assert gt.dtype == torch.uint8 and gt.is_contiguous() and gt.shape == (B, H, W)
        counts = counts_out if counts_out is not None else torch.empty(B, 4, dtype=torch.int64, device=x.device)
    L.check(L.lib().dfw_seg_postprocess_ex(x.data_ptr(), u8.data_ptr(), _p(gt), _p(counts), scratch.data_ptr(),
                                           B, H, W, float(r_threshold), float(threshold), int(bool(batch_max)),
                                           _stream()), "dfw_seg_postprocess")
    return u8, counts
