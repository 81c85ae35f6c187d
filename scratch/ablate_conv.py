"""Ablation of conv_patch_kernel's main loop (needs the library built with scratch/conv_patch_ablate.patch.txt; results of the
ablated runs are wrong on purpose).  mask bits: 1 no DMA issue, 2 no fragment reads, 4 no barriers, 8 no MFMAs, 16 no epilogue."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, _lib
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, Ci, Co) in [(12, 256, 256, 256), (12, 512, 128, 128)]:
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
    line = f"{B}x{H}^2 {Ci}->{Co}:"
    for mask in (0, 0, 1, 2, 4, 8, 16, 3, 7, 23, 9, 10, 12, 14, 30, 27, 29):
        _lib.configure(big_bk=100 + mask)
        line += f" [{mask:2d}] {t(lambda: ops.conv3x3(x, w, Co)):6.0f}"
    print(line, flush=True)
_lib.configure()
