import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
# lock-step launches as the UNet issues them: [4 support ; 4 query], 1-shot
for (heads, N) in [(5, 4096), (10, 1024), (20, 256), (20, 64)]:
    C = heads * 64
    qkv = torch.randn(8, N, 3 * C, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:]
    f = lambda: ops.fsa_attention(q, k, v, heads, k[:4], v[:4], nshot=1, n_plain=4)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); t = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 4.0 * heads * N * N * 64 * (4 + 8)
    print(f"pair h={heads} N={N}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
