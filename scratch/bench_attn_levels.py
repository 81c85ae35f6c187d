"""KV-fusion attention forward at the shapes the UNet launches it (lock-step batch [support ; query], q pre-scaled)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, _lib
dt = torch.bfloat16
for (b, nshot, heads, N) in [(4, 1, 5, 4096), (2, 5, 5, 4096), (1, 7, 5, 4096)]:
    C = heads * 64
    n_ref = b * nshot
    qkv = torch.randn(n_ref + b, N, 3 * C, device="cuda").to(dt)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    f = lambda: ops.fsa_attention(q, k, v, heads, k[:n_ref], v[:n_ref], nshot=nshot, n_plain=n_ref, q_prescaled=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); t = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 4.0 * heads * 64 * N * N * (n_ref + b * (1 + nshot))
    print(f"b={b} {nshot}-shot h={heads} N={N}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s  ", flush=True)
