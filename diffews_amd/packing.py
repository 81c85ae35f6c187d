"""Load-time weight repacking from the diffusers layout into the layouts the gfx950 kernels read.
Pure data movement (permute / concat / cast), done once per checkpoint on the host."""
import torch


def pack_conv3x3(w):
    """[Cout, Cin, 3, 3] -> [Cout, 9*Cin] with k = (ky*3 + kx)*Cin + c (implicit-GEMM K order)."""
    co, ci = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()


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
