"""Plain square GEMMs through dfw_gemm (gemm_big): python scratch/bench_sq.py -- compare with the guide's 256^2 8-phase template
(1320-1340 TF @4096^3, ~1470 @8192^3 on uniform random operands)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 1280, 1280), (32768, 512, 4608), (65536, 256, 2304)]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    tt = t(lambda: ops.linear(x, w))
    print(f"lin M={M} N={N} K={K}: {tt:8.1f} us {2.0*M*N*K/tt/1e6:7.1f} TF/s", flush=True)
    y = torch.matmul(x, w.t()); 
    tt = t(lambda: torch.matmul(x, w.t()))
    print(f"   torch.matmul (hipBLASLt)      : {tt:8.1f} us {2.0*M*N*K/tt/1e6:7.1f} TF/s", flush=True)
