"""Cross-attention (77 prompt tokens) forward / backward on the MFMA path at the training shapes: python scratch/bench_xattn_bwd.py"""
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
    return e0.elapsed_time(e1) / n * 1e3
B, L = 8, 77
tot_f = tot_b = 0.0
for (heads, N, layers) in [(5, 4096, 5), (10, 1024, 5), (20, 256, 5), (20, 64, 1)]:
    C = heads * 64
    q = (torch.randn(B, N, C, device="cuda") * 0.5).to(torch.bfloat16)
    kv = (torch.randn(B, L, 2 * C, device="cuda") * 0.5).to(torch.bfloat16)
    dout = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
    dkv = torch.empty_like(kv)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    fw = lambda: ops.fsa_attention(q, kv[..., :C], kv[..., C:], heads, q_prescaled=True, lse=lse)
    out = fw()
    tf = t(fw)
    tb = t(lambda: ops_bwd.attention_bwd(q, kv[..., :C], kv[..., C:], out, dout, lse, heads, dkv[..., :C], dkv[..., C:]))
    tot_f += tf * layers; tot_b += tb * layers
    print(f"heads={heads:2d} N={N:5d}: fwd {tf:7.1f} us | bwd {tb:7.1f} us  (x{layers} layers)", flush=True)
print(f"per step: fwd {tot_f/1e3:.2f} ms, bwd {tot_b/1e3:.2f} ms")
