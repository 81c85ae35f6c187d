"""Per-step vs fixed cost of the TN GEMM: same tile grid and split count, growing M."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops_bwd
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Ci, Co) in [(320, 320), (1280, 1280), (1280, 5120)]:
    for M in (4096, 32768, 65536, 131072, 262144):
        dy = torch.randn(M, Co, device="cuda").to(torch.bfloat16)
        x = torch.randn(M, Ci, device="cuda").to(torch.bfloat16)
        out = torch.zeros(1, Co, 1, Ci, device="cuda")
        ms = t(lambda: ops_bwd.gemm_tn(dy, x, out=out, accumulate=True))
        fl = 2.0 * M * Co * Ci
        print(f"{Ci}->{Co} M={M:7d}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s", flush=True)
