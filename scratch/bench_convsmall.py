"""conv_small (VAE conv_in) timing: python scratch/bench_convsmall.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, packing

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (B, Cin, H, Cout) in [(12, 3, 512, 128), (4, 4, 64, 512), (8, 8, 64, 320)]:
    x = torch.randn(B, Cin, H, H, device="cuda")
    w = packing.pack_conv_small(torch.randn(Cout, Cin, 3, 3) * 0.1).cuda()
    b = torch.randn(Cout, device="cuda")
    ms = t(lambda: ops.conv_small(x, w, b, Cout, 9, torch.bfloat16))
    nb = B * H * H * Cout * 2 + x.numel() * 4
    print(f"B{B} {Cin}->{Cout} @{H}: {ms*1e3:.0f} us = {nb/ms/1e9:.2f} TB/s")
