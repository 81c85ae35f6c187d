"""Ablation of gemm_big's main loop on a plain GEMM: needs the library built with scratch/gemm_big_ablate.patch.txt applied
mask bits: 1 no DMA issue, 2 no fragment reads, 4 no barriers, 8 no MFMAs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, _lib
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(4096, 4096, 4096), (32768, 512, 4608)]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    line = f"M={M} N={N} K={K}:"
    for mask in (0, 0, 1, 2, 4, 8, 3, 5, 6, 7, 9, 10, 12, 14, 11, 13):
        _lib.configure(big_bk=100 + mask)
        tt = t(lambda: ops.linear(x, w))
        line += f" [{mask:2d}] {tt:6.1f}"
    print(line, flush=True)
_lib.configure()
