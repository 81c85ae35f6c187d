"""Self-attention backward at the training shapes: python scratch/bench_attn_bwd.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, ops_bwd
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, heads, N, nshot, n_plain) in [(8, 5, 4096, 7, 7), (8, 10, 1024, 7, 7), (8, 20, 256, 7, 7), (8, 5, 4096, 0, 0)]:
    C = heads * 64
    qkv = (torch.randn(B, N, 3 * C, device="cuda") * 0.5).to(torch.bfloat16)
    dout = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    if nshot:
        fw = lambda: ops.fsa_attention(q, k, v, heads, k[:n_plain], v[:n_plain], nshot=nshot, n_plain=n_plain, q_prescaled=True, lse=lse)
    else:
        fw = lambda: ops.fsa_attention(q, k, v, heads, q_prescaled=True, lse=lse)
    out = fw()
    tf = t(fw)
    tb = t(lambda: ops_bwd.fsa_attention_bwd(qkv, out, dout, lse, heads, nshot=nshot, n_plain=n_plain))
    keys = N * (B if not nshot else n_plain + (B - n_plain) * (1 + nshot))
    fl = 4.0 * heads * N * keys * 64
    print(f"B={B} h={heads} N={N} nshot={nshot}: fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.1f} TF/s | bwd {tb*1e3:7.1f} us {2.5*fl/tb/1e9:6.1f} TF/s (ratio {tb/tf:.2f})", flush=True)
