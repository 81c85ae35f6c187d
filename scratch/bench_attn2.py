import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
for (B, heads, N, nshot) in [(4, 4, 4096, 0), (8, 4, 4096, 0), (16, 4, 4096, 0), (4, 5, 4096, 0), (8, 5, 4096, 0), (4, 4, 4096, 1), (8, 4, 4096, 1)]:
    C = heads * 64
    qkv = torch.randn(B, N, 3 * C, device="cuda").to(torch.bfloat16)
    bank = torch.randn(max(1, B * nshot), N, 3 * C, device="cuda").to(torch.bfloat16)
    f = lambda: ops.fsa_attention(qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:], heads,
                                  bank[..., C:2*C] if nshot else None, bank[..., 2*C:] if nshot else None, nshot=nshot)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); t = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 4.0 * B * heads * N * N * (1 + nshot) * 64
    print(f"B={B} h={heads} N={N} nshot={nshot}: WGs={B*heads*N//256:5d} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
