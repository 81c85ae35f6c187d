"""gemm_big on plain square GEMMs (compare with the guide's 256^2 8-phase template: 1320-1470 TF random data)."""
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
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (65536, 512, 4608), (65536, 256, 2304)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    ms = t(lambda: ops.linear(a, w))
    print(f"{M}x{N}x{K}: {ms*1e3:8.0f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
    az, wz = torch.zeros_like(a), torch.zeros_like(w)
    ms = t(lambda: ops.linear(az, wz))
    print(f"   zero-filled: {ms*1e3:8.0f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
