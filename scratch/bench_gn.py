"""GroupNorm(+SiLU) bandwidth on the VAE shapes: python scratch/bench_gn.py"""
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
    return e0.elapsed_time(e1) / n

for (B, H, C) in [(9, 512, 128), (9, 256, 256), (9, 128, 512), (4, 512, 128), (4, 256, 256), (4, 64, 512)]:
    x = torch.randn(B, H, H, C, device="cuda", dtype=torch.bfloat16)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ms = t(lambda: ops.groupnorm(x, g, b, 32, 1e-6, silu=True))
    y = torch.empty_like(x)
    mc = t(lambda: y.copy_(x))
    nb = x.numel() * 2
    print(f"B{B} {H}x{H}x{C}: gn(stats+apply) {ms*1e3:.0f} us = {3*nb/ms/1e9:.2f} TB/s (3 passes of {nb/1e6:.0f} MB); torch copy {mc*1e3:.0f} us = {2*nb/mc/1e9:.2f} TB/s")
