"""Key-split count of the bank readers (forward and dQ), 7-shot lock-step launches: python scratch/sweep_splits.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, ops_bwd, _lib
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, heads, N, nshot, n_plain) in [(8, 5, 4096, 7, 7), (8, 10, 1024, 7, 7), (8, 20, 256, 7, 7)]:
    C = heads * 64
    qkv = (torch.randn(B, N, 3 * C, device="cuda") * 0.5).to(torch.bfloat16)
    dout = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    line = f"h={heads} N={N}:"
    for fs in (4, 0, 4, 0, 3, 5):
        _lib.configure(fsa_force_splits=fs) if fs != 1 else _lib.configure(fsa_key_split=0)
        fw = lambda: ops.fsa_attention(q, k, v, heads, k[:n_plain], v[:n_plain], nshot=nshot, n_plain=n_plain, q_prescaled=True, lse=lse)
        out = fw()
        tf = t(fw)
        tb = t(lambda: ops_bwd.fsa_attention_bwd(qkv, out, dout, lse, heads, nshot=nshot, n_plain=n_plain))
        line += f"  splits {fs if fs else 'auto'}: fwd {tf:6.1f} bwd {tb:7.1f} |"
        _lib.configure()
    print(line, flush=True)
