"""Host-side wrappers of the backward kernels (include/diffews_hip.h, "Training step").

Counterpart of ops.py for the training step of
/root/reference/train_tools/train_icl_multitask_nocrop_nearest_nshot_v3.py:1374-1396: torch tensors carry device
memory and the current stream only; every gradient is computed by libdiffews_hip.so.  Activation gradients are
NHWC / [rows, C] tensors in the engine storage dtype, parameter gradients fp32.
"""
import ctypes as C

import torch

from . import _lib as L
from .ops import _dt, _p, _stream, _f32


def _ws(nbytes, device):
    return torch.empty(max(1, (nbytes + 3) // 4), dtype=torch.float32, device=device)


def gemm_tn(dy, x, n=None, kc=None, out=None, taps=1, geom=None, accumulate=False, scale=1.0, batch=1, batch2=1,
            strides=(0, 0, 0, 0), M=None, lda=None, ldb=None, out_strides=(0, 0, 0)):
    """out[b][n][tap][k] (+)= scale * sum_m dy[m][n] * x[row(m, tap)][k]   (fp32).
    dy [M, >=n] / x [rows, >=kc] are 2-D (or NHWC) storage-dtype tensors whose last dim is contiguous.
    geom = (Hi, Wi, Ho, Wo, stride, pad, ups) for taps == 9 (the forward conv's geometry)."""
    dy2 = dy.reshape(-1, dy.shape[-1]) if dy.dim() != 2 else dy
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
    assert dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.dtype == x2.dtype
    M = M if M is not None else dy2.shape[0]
    n = n if n is not None else dy2.shape[1]
    kc = kc if kc is not None else x2.shape[1]
    nb = max(1, batch) * max(1, batch2)
    if out is None:
        out = torch.empty(nb, n, taps, kc, dtype=torch.float32, device=dy.device)
        assert not accumulate
    a = L.GemmTnArgs()
    a.A, a.B, a.out = dy2.data_ptr(), x2.data_ptr(), out.data_ptr()
    a.a_elems = dy2.numel() if dy2.is_contiguous() else (dy2.shape[0] - 1) * dy2.stride(0) + dy2.shape[1]
    a.b_elems = x2.numel() if x2.is_contiguous() else (x2.shape[0] - 1) * x2.stride(0) + x2.shape[1]
    if nb > 1:   # batched problems address beyond one matrix: the extents are those of the whole buffers
        a.a_elems = dy.numel() if dy.is_contiguous() else a.a_elems
        a.b_elems = x.numel() if x.is_contiguous() else a.b_elems
    a.M, a.N, a.Kc = M, n, kc
    a.lda = lda if lda is not None else dy2.stride(0)
    a.ldb = ldb if ldb is not None else x2.stride(0)
    a.taps = taps
    if taps == 9:
        a.Hi, a.Wi, a.Ho, a.Wo, a.stride, a.pad, a.ups = [int(v) for v in geom]
    a.batch, a.batch2 = batch, batch2
    a.strideA, a.strideB, a.strideA2, a.strideB2 = [int(v) for v in strides]
    a.ldo_n, a.ldo_t, a.ldo_b = [int(v) for v in out_strides]
    a.scale, a.accumulate, a.dtype = scale, int(accumulate), _dt(dy2)
    lib = L.lib()
    nbytes = lib.dfw_gemm_tn_workspace_bytes(C.byref(a))
    if nbytes == 0:
        L.check(lib.dfw_gemm_tn(C.byref(a), _stream()), "dfw_gemm_tn")   # reports the argument error
    ws = _ws(nbytes, dy.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    L.check(lib.dfw_gemm_tn(C.byref(a), _stream()), "dfw_gemm_tn")
    return out


def colsum(x, segs=1, out=None, accumulate=False, scale=1.0):
    """Column sums of x [rows, N] (storage dtype) per segment of rows / segs rows -> fp32 [segs, N]."""
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.stride(1) == 1
    rows, N = x2.shape
    assert rows % segs == 0
    if out is None:
        out = torch.empty(segs, N, dtype=torch.float32, device=x.device)
        assert not accumulate
    ldo = out.stride(0) if out.dim() == 2 else N
    lib = L.lib()
    nbytes = lib.dfw_colsum_workspace_bytes(rows // segs, segs, N)
    ws = _ws(nbytes, x.device)
    L.check(lib.dfw_colsum(x2.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, rows // segs, segs, N, x2.stride(0), ldo,
                           float(scale), int(accumulate), _dt(x2), _stream()), "dfw_colsum")
    return out


class ColsumQueue:
    """Deferred column sums of a backward walk: add() records what colsum() would launch (and keeps the operand alive),
    flush() issues everything recorded so far in TWO launches (dfw_colsum_batch) instead of two per item -- ~190 bias /
    time-projection gradients per training step.  Results are identical to colsum()'s (same partial-sum plan per item, same
    fold order).  flush() must run before anything reads an output (the trainer flushes before the time-projection closure,
    before it reports gradients as final to the gradient reducer, and at the end of the walk)."""

    def __init__(self):
        self.items, self.keep = [], []

    def add(self, x, out, segs=1, accumulate=False, scale=1.0):
        x2 = x.reshape(-1, x.shape[-1])
        assert x2.stride(1) == 1 and out.dtype == torch.float32
        rows, N = x2.shape
        assert rows % segs == 0 and N % 8 == 0 and x2.stride(0) % 8 == 0
        ldo = out.stride(0) if out.dim() == 2 else N
        if self.items:
            assert self.items[0][-1] == _dt(x2), "one storage dtype per batch"
        self.items.append((x2.data_ptr(), out.data_ptr(), rows // segs, segs, N, x2.stride(0), ldo, float(scale), int(accumulate), _dt(x2)))
        self.keep.append((x2, out))

    def flush(self):
        if not self.items:
            return
        import struct
        lib = L.lib()
        rows, b1, b2, off = [], 0, 0, 0
        ch, rp = C.c_int32(0), C.c_int32(0)
        for (xp, op, rps, segs, N, ldx, ldo, scale, acc, _) in self.items:
            L.check(lib.dfw_colsum_plan(rps, C.byref(ch), C.byref(rp)), "dfw_colsum_plan")
            sbits = struct.unpack("<i", struct.pack("<f", scale))[0]
            rows.append([xp, op, rps, segs, N, ldx, ldo, sbits, acc, ch.value, rp.value, off, b1, b2, 0, 0])
            b1 += ((N + 255) // 256) * ch.value * segs
            b2 += ((N + 15) // 16) * segs
            off += segs * ch.value * N
        dev = self.keep[0][0].device
        table = torch.empty(len(rows), 16, dtype=torch.int64, device=dev)
        for i0 in range(0, len(rows), 24):          # the records travel as kernel arguments (capture-safe: no pinned memory)
            chunk = rows[i0:i0 + 24]
            flat = (C.c_int64 * (16 * len(chunk)))(*[v for r in chunk for v in r])
            L.check(lib.dfw_table_write(table.data_ptr(), i0, flat, len(chunk), _stream()), "dfw_table_write")
        ws = torch.empty(max(off, 1), dtype=torch.float32, device=dev)
        L.check(lib.dfw_colsum_batch(table.data_ptr(), len(rows), b1, b2, ws.data_ptr(), self.items[0][-1], _stream()), "dfw_colsum_batch")
        self.items, self.keep = [], []


def groupnorm_bwd(x, dy, mean_rstd, gamma, beta, groups, silu, dgamma=None, dbeta=None, accumulate=False, grad_scale=1.0, dx_add=None):
    """-> dx (like x).  x, dy NHWC (or [B, HW, C]) contiguous; mean_rstd [B, groups, 2] fp32 from the forward.
    dx_add: a gradient x already has (same memory layout as x), added in the kernel (fp32 sum, one rounding)."""
    assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape and x.dtype == dy.dtype
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    dx = torch.empty_like(x)
    a = L.GroupNormBwdArgs()
    a.x, a.dy, a.dx = x.data_ptr(), dy.data_ptr(), dx.data_ptr()
    a.gamma, a.beta, a.mean_rstd = _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta")), mean_rstd.data_ptr()
    a.dgamma, a.dbeta = _p(dgamma), _p(dbeta)
    a.B, a.HW, a.C, a.groups, a.ldx, a.lddy, a.lddx = B, HW, Cc, groups, Cc, Cc, Cc
    a.silu, a.accumulate, a.grad_scale, a.dtype = int(silu), int(accumulate), grad_scale, _dt(x)
    if dx_add is not None:
        assert dx_add.is_contiguous() and dx_add.numel() == x.numel() and dx_add.dtype == x.dtype
        a.dx_add = dx_add.data_ptr()
    lib = L.lib()
    nbytes = lib.dfw_groupnorm_bwd_workspace_bytes(C.byref(a))
    ws = _ws(nbytes, x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    L.check(lib.dfw_groupnorm_bwd(C.byref(a), _stream()), "dfw_groupnorm_bwd")
    return dx


def layernorm_bwd(x, dy, gamma, dgamma, dbeta, eps=1e-5, accumulate=False, grad_scale=1.0, dx_add=None):
    assert x.dim() == 2 and dy.shape == x.shape and x.stride(1) == 1 and dy.stride(1) == 1
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    a = L.LayerNormBwdArgs()
    a.x, a.dy, a.dx, a.gamma = x.data_ptr(), dy.data_ptr(), dx.data_ptr(), _p(_f32(gamma, "gamma"))
    a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    a.rows, a.C, a.ldx, a.lddy, a.lddx = x.shape[0], x.shape[1], x.stride(0), dy.stride(0), dx.stride(0)
    a.eps, a.accumulate, a.grad_scale, a.dtype = eps, int(accumulate), grad_scale, _dt(x)
    if dx_add is not None:     # rows at dx's stride (dx is contiguous)
        assert dx_add.is_contiguous() and dx_add.numel() == x.numel() and dx_add.dtype == x.dtype
        a.dx_add = dx_add.data_ptr()
    lib = L.lib()
    nbytes = lib.dfw_layernorm_bwd_workspace_bytes(x.shape[0], x.shape[1])
    ws = _ws(nbytes, x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    L.check(lib.dfw_layernorm_bwd(C.byref(a), _stream()), "dfw_layernorm_bwd")
    return dx


def geglu_fwd(pre):
    """pre [rows, 2H] in the packed column order (packing.pack_geglu) -> [rows, H]."""
    assert pre.dim() == 2 and pre.is_contiguous()
    rows, H2 = pre.shape
    out = torch.empty(rows, H2 // 2, dtype=pre.dtype, device=pre.device)
    L.check(L.lib().dfw_geglu(pre.data_ptr(), None, out.data_ptr(), rows, H2 // 2, _dt(pre), _stream()), "dfw_geglu")
    return out


def geglu_bwd(pre, dout):
    assert pre.is_contiguous() and dout.is_contiguous() and dout.shape == (pre.shape[0], pre.shape[1] // 2)
    dpre = torch.empty_like(pre)
    L.check(L.lib().dfw_geglu(pre.data_ptr(), dout.data_ptr(), dpre.data_ptr(), pre.shape[0], pre.shape[1] // 2, _dt(pre),
                              _stream()), "dfw_geglu")
    return dpre


def _ew(mode, a, b, y, rows, Cc, lda=0, c0=0, H=0, W=0):
    L.check(L.lib().dfw_elementwise(mode, a.data_ptr(), _p(b), y.data_ptr(), rows, Cc, lda, c0, H, W, _dt(a), _stream()),
            "dfw_elementwise")
    return y


def add(a, b):
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    Cc = a.shape[-1]
    return _ew(0, a, b, torch.empty_like(a), a.numel() // Cc, Cc)


def slice_channels(a, c0, Cc):
    """a[..., c0:c0+Cc] as a contiguous tensor (backward of concat_channels)."""
    assert a.is_contiguous()
    y = torch.empty(*a.shape[:-1], Cc, dtype=a.dtype, device=a.device)
    return _ew(1, a, None, y, a.numel() // a.shape[-1], Cc, lda=a.shape[-1], c0=c0)


def zero_stuff2x(a):
    """[B, H, W, C] -> [B, 2H, 2W, C] with a at the even positions, zeros elsewhere."""
    assert a.is_contiguous() and a.dim() == 4
    B, H, W, Cc = a.shape
    y = torch.empty(B, 2 * H, 2 * W, Cc, dtype=a.dtype, device=a.device)
    return _ew(2, a, None, y, B * 4 * H * W, Cc, H=H, W=W)


def pool2x2_sum(a):
    """[B, 2H, 2W, C] -> [B, H, W, C], sums of the 2x2 blocks (backward of the nearest-2x upsample)."""
    assert a.is_contiguous() and a.dim() == 4
    B, H2, W2, Cc = a.shape
    y = torch.empty(B, H2 // 2, W2 // 2, Cc, dtype=a.dtype, device=a.device)
    return _ew(3, a, None, y, B * (H2 // 2) * (W2 // 2), Cc, H=H2 // 2, W=W2 // 2)


def nchw_to_nhwc(x, dtype, cp=8, scale=1.0):
    """NCHW fp32 -> NHWC `dtype` with channels zero-padded to cp."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, cp, dtype=dtype, device=x.device)
    L.check(L.lib().dfw_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), B, Cc, H * W, cp, float(scale), L.BF16 if dtype == torch.bfloat16 else L.F16,
                                     _stream()), "dfw_nchw_to_nhwc")
    return y


def mse_loss(pred, target, dtype, loss_scale=1.0, dpred_out=None, dpred_nchw_out=None):
    """-> (loss fp32 tensor [1], dpred NHWC [B, H, W, 8] in `dtype` = 2 (pred - target) / numel * loss_scale).
    dpred_out: zero-initialised destination (e.g. the query rows of a larger batch); dpred_nchw_out: optional fp32
    [B, C, H, W] that receives the same rounded values."""
    assert pred.shape == target.shape and pred.dtype == torch.float32 and pred.is_contiguous() and target.is_contiguous()
    B, Cc, H, W = pred.shape
    dpred = dpred_out if dpred_out is not None else torch.zeros(B, H, W, 8, dtype=dtype, device=pred.device)
    assert dpred.shape == (B, H, W, 8) and dpred.is_contiguous() and dpred.dtype == dtype
    if dpred_nchw_out is not None:
        assert dpred_nchw_out.shape == pred.shape and dpred_nchw_out.is_contiguous() and dpred_nchw_out.dtype == torch.float32
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    ws = torch.empty(256, dtype=torch.float32, device=pred.device)
    L.check(L.lib().dfw_mse_loss(pred.data_ptr(), target.data_ptr(), dpred.data_ptr(), _p(dpred_nchw_out), loss.data_ptr(),
                                 ws.data_ptr(), B, Cc, H * W, float(loss_scale), L.BF16 if dtype == torch.bfloat16 else L.F16,
                                 _stream()), "dfw_mse_loss")
    return loss, dpred


def fsa_attention_bwd(qkv, out, dout, lse, heads, nshot=0, n_plain=0, scale=None, key_split=True):
    """qkv [B, N, 3C] (q pre-scaled), out / dout [B, N, C], lse [B, heads, N] -> dqkv [B, N, 3C]."""
    B, N, C3 = qkv.shape
    Cq = heads * 64
    assert C3 == 3 * Cq and qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous()
    assert out.shape == (B, N, Cq) and dout.shape == out.shape and lse.shape == (B, heads, N) and lse.dtype == torch.float32
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(2, B, heads, N, dtype=torch.float32, device=qkv.device)   # (-delta | -lse), see the header
    a = L.FsaBwdArgs()
    a.qkv, a.out, a.dout, a.lse, a.delta, a.dqkv = qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr()
    a.delta_bytes = delta.numel() * 4
    a.batch, a.heads, a.n, a.nshot, a.n_plain = B, heads, N, nshot, n_plain
    a.ld, a.ldo, a.ldd = C3, Cq, C3
    a.scale = scale if scale is not None else 64 ** -0.5
    a.dtype = _dt(qkv)
    lib = L.lib()
    nbytes = lib.dfw_fsa_attention_bwd_workspace_bytes(C.byref(a)) if (nshot >= 2 and key_split) else 0
    if nbytes:      # dQ key split of the bank readers (many shots)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=qkv.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    from . import ops as _ops
    if _ops.gemm_hook is not None:      # bench.py roofline leg: the dQ + dK/dV kernels of this launch (5 GEMMs = 2.5 x the forward's 2)
        keys = n_plain * N + (B - n_plain) * (N + nshot * N)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dfw_fsa_attention_bwd(C.byref(a), _stream()), "dfw_fsa_attention_bwd")
        e1.record()
        _ops.gemm_hook("fsa_attention_bwd", 10.0 * heads * 64 * N * keys, e0, e1, (B, heads, N, keys))
        return dqkv
    L.check(lib.dfw_fsa_attention_bwd(C.byref(a), _stream()), "dfw_fsa_attention_bwd")
    return dqkv


def attention_bwd(q, k, v, out, dout, lse, heads, dk_out, dv_out, scale=None, q_split=True):
    """Flash backward with queries and keys / values in their own tensors (attn2 on the MFMA path).  q [B, N, heads*64]
    PRE-SCALED (linear(..., colscale=(C, FSA_QSCALE))); k / v [B, L, heads*64] column slices of one buffer; out / dout
    [B, N, heads*64] contiguous; lse [B, heads, N] from fsa_attention(..., lse=); dk_out / dv_out: views that receive
    dK / dV.  -> dq [B, N, heads*64] with respect to the unscaled projection output."""
    B, N, Cq = q.shape
    Lc = k.shape[1]
    assert Cq == heads * 64 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert k.stride(0) == v.stride(0) and k.stride(1) == v.stride(1) and v.data_ptr() >= k.data_ptr()
    assert out.shape == (B, N, Cq) and dout.shape == out.shape and out.is_contiguous() and dout.is_contiguous()
    assert lse.shape == (B, heads, N) and lse.dtype == torch.float32 and lse.is_contiguous()
    assert dk_out.shape == (B, Lc, Cq) and dv_out.shape == dk_out.shape and dk_out.stride(2) == 1
    assert dk_out.stride(0) == dv_out.stride(0) and dk_out.stride(1) == dv_out.stride(1)
    dq = torch.empty(B, N, Cq, dtype=q.dtype, device=q.device)
    delta = torch.empty(2, B, heads, N, dtype=torch.float32, device=q.device)
    a = L.AttnBwdArgs()
    a.q, a.k, a.v, a.out, a.dout, a.lse, a.delta = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr()
    a.delta_bytes = delta.numel() * 4
    a.dq, a.dk, a.dv = dq.data_ptr(), dk_out.data_ptr(), dv_out.data_ptr()
    a.batch, a.heads, a.n_q, a.n_kv = B, heads, N, Lc
    a.ldq, a.ldkv, a.ldo, a.lddq, a.lddkv = q.stride(1), k.stride(1), Cq, Cq, dk_out.stride(1)
    a.q_bs, a.kv_bs, a.o_bs, a.dq_bs, a.dkv_bs = q.stride(0), k.stride(0), N * Cq, N * Cq, dk_out.stride(0)
    a.scale = scale if scale is not None else 64 ** -0.5
    a.dtype = _dt(q)
    nbytes = L.lib().dfw_attention_bwd_workspace_bytes(C.byref(a)) if q_split else 0
    if nbytes:      # short key axis: query split of the dK/dV kernel
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    L.check(L.lib().dfw_attention_bwd(C.byref(a), _stream()), "dfw_attention_bwd")
    return dq


def cross_attention_bwd(q, k, v, dout, heads, dk_out, dv_out, scale=None):
    """attn2 backward.  q / dout [B, N, heads*64]; k / v [B, L, heads*64] views; dk_out / dv_out: [B, L, heads*64]
    views (column slices of the fused prompt-K/V gradient buffer) that receive dK / dV.  -> dq [B, N, heads*64]."""
    B, N, Cq = q.shape
    Lc = k.shape[1]
    assert Cq == heads * 64 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1 and dout.stride(2) == 1
    assert dk_out.shape == (B, Lc, Cq) and dv_out.shape == dk_out.shape and dk_out.stride(2) == 1
    assert dk_out.stride(0) == dv_out.stride(0) and dk_out.stride(1) == dv_out.stride(1)
    dq = torch.empty(B, N, Cq, dtype=q.dtype, device=q.device)
    a = L.XattnBwdArgs()
    a.q, a.k, a.v, a.dout, a.dq, a.dk, a.dv = q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr(), dq.data_ptr(), dk_out.data_ptr(), dv_out.data_ptr()
    a.batch, a.heads, a.n_q, a.L = B, heads, N, Lc
    a.ldq, a.ldk, a.ldv, a.ldo, a.lddq, a.lddkv = q.stride(1), k.stride(1), v.stride(1), dout.stride(1), Cq, dk_out.stride(1)
    a.q_bs, a.k_bs, a.v_bs, a.o_bs, a.dq_bs, a.dkv_bs = q.stride(0), k.stride(0), v.stride(0), dout.stride(0), N * Cq, dk_out.stride(0)
    a.scale = scale if scale is not None else 64 ** -0.5
    a.dtype = _dt(q)
    lib = L.lib()
    nbytes = lib.dfw_cross_attention_bwd_workspace_bytes(B, heads, N, Lc)
    ws = _ws(nbytes, q.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    L.check(lib.dfw_cross_attention_bwd(C.byref(a), _stream()), "dfw_cross_attention_bwd")
    return dq


def silu(a, dy=None):
    """silu(a), or dy * silu'(a) when dy is given (storage dtype, any shape, contiguous)."""
    assert a.is_contiguous() and (dy is None or (dy.is_contiguous() and dy.shape == a.shape))
    y = torch.empty_like(a)
    L.check(L.lib().dfw_silu(a.data_ptr(), _p(dy), y.data_ptr(), a.numel(), _dt(a), _stream()), "dfw_silu")
    return y


def sumsq(x):
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    L.check(L.lib().dfw_sumsq(x.data_ptr(), out.data_ptr(), ws.data_ptr(), x.numel(), _stream()), "dfw_sumsq")
    return out


def linear_wt(w):
    """W [N, K] (storage dtype, contiguous) -> W^T [K, N]: the weight of the data-gradient GEMM dX = dY W."""
    assert w.dim() == 2 and w.is_contiguous()
    N, K = w.shape
    y = torch.empty(K, N, dtype=w.dtype, device=w.device)
    L.check(L.lib().dfw_weight_relayout(w.data_ptr(), y.data_ptr(), N, K, K, N, 1, 0, 0, 0, _stream()), "dfw_weight_relayout")
    return y


def conv3x3_wd(w, cout):
    """packed conv weight [Cout, 9*Cin] -> [Cin, 9*Cout] with mirrored taps (packing.pack_conv3x3_dgrad on device)."""
    assert w.dim() == 2 and w.is_contiguous() and w.shape[0] == cout
    cin = w.shape[1] // 9
    y = torch.empty(cin, 9 * cout, dtype=w.dtype, device=w.device)
    L.check(L.lib().dfw_weight_relayout(w.data_ptr(), y.data_ptr(), cout, cin, 9 * cin, 9 * cout, 9, cin, cout, 1, _stream()),
            "dfw_weight_relayout")
    return y


def relayout_table(entries, device):
    """entries: [(kind, w, y)] with kind 'T' (Linear W [N, K] -> y [K, N]) or 'D' (packed conv W [Cout, 9*Cin] ->
    y [Cin, 9*Cout], taps mirrored).  -> (device int64 table [n, 12], n, total_blocks) for weight_relayout_batch: the
    tensors' addresses are baked in, so the table is valid while they live."""
    rows, blk = [], 0
    for kind, w, y in entries:
        assert w.is_contiguous() and y.is_contiguous() and w.dtype == y.dtype
        if kind == "T":
            N, K = w.shape
            R, Cc, ldx, ldy, nb, x_bs, y_bs, flip = N, K, K, N, 1, 0, 0, 0
        else:
            cout, cin = w.shape[0], w.shape[1] // 9
            R, Cc, ldx, ldy, nb, x_bs, y_bs, flip = cout, cin, 9 * cin, 9 * cout, 9, cin, cout, 1
        assert R % 8 == 0 and Cc % 8 == 0 and w.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0
        rows.append([w.data_ptr(), y.data_ptr(), R, Cc, ldx, ldy, x_bs, y_bs, nb, flip, blk, 0])
        blk += ((Cc + 63) // 64) * ((R + 63) // 64) * nb
    return torch.tensor(rows, dtype=torch.int64).to(device), len(rows), blk


def weight_relayout_batch(table):
    t, n, blocks = table
    L.check(L.lib().dfw_weight_relayout_batch(t.data_ptr(), n, blocks, _stream()), "dfw_weight_relayout_batch")


def loss_grad(g, dtype, scale=1.0, dpred_out=None, dpred_nchw_out=None):
    """External d loss / d pred (NCHW fp32 [B, C, H, W]) -> the same two tensors mse_loss hands the backward:
    dpred NHWC [B, H, W, 8] in `dtype` = round(g * scale) (channels C..7 untouched) and its NCHW fp32 copy."""
    assert g.dtype == torch.float32 and g.is_contiguous() and g.dim() == 4
    B, Cc, H, W = g.shape
    dpred = dpred_out if dpred_out is not None else torch.zeros(B, H, W, 8, dtype=dtype, device=g.device)
    assert dpred.shape == (B, H, W, 8) and dpred.is_contiguous() and dpred.dtype == dtype
    if dpred_nchw_out is not None:
        assert dpred_nchw_out.shape == g.shape and dpred_nchw_out.is_contiguous() and dpred_nchw_out.dtype == torch.float32
    L.check(L.lib().dfw_loss_grad(g.data_ptr(), dpred.data_ptr(), _p(dpred_nchw_out), B, Cc, H * W, float(scale),
                                  L.BF16 if dtype == torch.bfloat16 else L.F16, _stream()), "dfw_loss_grad")
    return dpred


def to_f32(x, out, scale=1.0):
    """out (fp32) = x (storage dtype) * scale, flat contiguous buffers."""
    assert out.dtype == torch.float32 and out.is_contiguous() and x.is_contiguous() and x.numel() == out.numel()
    L.check(L.lib().dfw_convert_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), float(scale), _dt(x), _stream()), "dfw_convert_to_f32")
    return out


def adamw(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_sumsq=None,
          max_grad_norm=0.0, shadow=None, found_inf=None):
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == param.numel()
    a = L.AdamWArgs()
    a.param, a.grad, a.exp_avg, a.exp_avg_sq = param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()
    a.grad_sumsq = _p(grad_sumsq)
    a.n, a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = param.numel(), lr, betas[0], betas[1], eps, weight_decay
    a.max_grad_norm, a.step = max_grad_norm, step
    if found_inf is not None:
        assert found_inf.dtype == torch.int32 and found_inf.numel() >= 1
        a.found_inf = found_inf.data_ptr()
    if shadow is not None:
        assert shadow.numel() == param.numel() and shadow.is_contiguous()
        a.shadow, a.shadow_dtype = shadow.data_ptr(), _dt(shadow)
    L.check(L.lib().dfw_adamw(C.byref(a), _stream()), "dfw_adamw")
