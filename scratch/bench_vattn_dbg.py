import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, N = 8, 4096
qkv = (torch.randn(B, N, 1536, device="cuda")).to(torch.bfloat16)
q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
for dbg, name in [(0, "all"), (1, "no DMA"), (0, "all"), (1, "no DMA")]:
    os.environ["DFW_VATTN_DBG"] = str(dbg)
    print(f"{name:22s} {t(lambda: ops.vae_attention(q, k, v)):8.1f} us", flush=True)
