"""Load-time weight repacking from the diffusers layout into the layouts the gfx950 kernels read.
Pure data movement (permute / concat / cast), done once per checkpoint on the host."""
import torch


def pack_conv3x3(w):
    """[Cout, Cin, 3, 3] -> [Cout, 9*Cin] with k = (ky*3 + kx)*Cin + c (implicit-GEMM K order)."""
    co, ci = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()


def block_conv3x3(wp):
    """packed [Cout, 9*Cin] -> blocked [9][Cin/32][Cout][32] (dfw_gemm_args.W_blocked): one (tap, 32-channel chunk)
    stage of the patch conv kernel is one contiguous Cout x 64-byte block."""
    co, k = wp.shape
    ci = k // 9
    assert k == 9 * ci and ci % 32 == 0
    return wp.view(co, 9, ci // 32, 32).permute(1, 2, 0, 3).contiguous()


def pack_conv1x1(w):
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def pack_conv_small(w):
    """[Cout, Cin, k, k] -> fp32 [Cout, k*k, Cin] for the boundary convs (Cin <= 8)."""
    co, ci, k, _ = w.shape
    return w.float().permute(0, 2, 3, 1).reshape(co, k * k, ci).contiguous()


def geglu_perm(n_half):
    """Row permutation for the fused GEGLU epilogue: packed rows come in 64-row groups,
    32 value rows followed by their 32 gate rows (value row j pairs with gate row n_half + j)."""
    assert n_half % 32 == 0
    t = torch.arange(n_half // 32)
    i = torch.arange(32)
    val = (t[:, None] * 32 + i[None, :])            # [T, 32]
    gate = val + n_half
    return torch.stack([val, gate], dim=1).reshape(-1)  # [T, 2, 32] flattened


def pack_geglu(w, b):
    """GEGLU proj weight [2*H, C] / bias [2*H] -> row-interleaved copies (see geglu_perm)."""
    perm = geglu_perm(w.shape[0] // 2)
    return w[perm].contiguous(), b.float()[perm].contiguous()


def pack_conv3x3_dgrad(w):
    """[Cout, Cin, 3, 3] -> [Cin, 9*Cout]: the weight of the conv that computes the DATA gradient of a stride-1
    conv3x3 (dX = conv3x3(dY, W') with W'[ci][ky][kx][co] = W[co][ci][2-ky][2-kx]: taps mirrored, channels swapped)."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, 9 * co).contiguous()


def unpack_conv3x3_grad(g):
    """packed gradient [Cout, 9, Cin] (dfw_gemm_tn, taps = 9) -> diffusers layout [Cout, Cin, 3, 3]."""
    co, _, ci = g.shape
    return g.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
