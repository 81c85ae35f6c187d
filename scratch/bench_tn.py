"""Weight-gradient (TN) GEMM on the UNet's layer shapes at the training batch (8 latents of 64x64):
python scratch/bench_tn.py"""
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

# (B, H, Cin, Cout, taps, count per step)
shapes = [(8, 64, 320, 320, 9, 7), (8, 64, 640, 320, 9, 2), (8, 64, 960, 320, 9, 1), (8, 32, 640, 640, 9, 6), (8, 32, 320, 640, 9, 1),
          (8, 32, 1280, 640, 9, 1), (8, 32, 960, 640, 9, 1), (8, 16, 1280, 1280, 9, 6), (8, 16, 640, 1280, 9, 1),
          (8, 16, 2560, 1280, 9, 2), (8, 16, 1920, 1280, 9, 1), (8, 8, 1280, 1280, 9, 7), (8, 8, 2560, 1280, 9, 3),
          (8, 64, 320, 320, 1, 25), (8, 64, 320, 960, 1, 5), (8, 64, 320, 2560, 1, 5), (8, 64, 1280, 320, 1, 5),
          (8, 32, 640, 640, 1, 25), (8, 32, 640, 1920, 1, 5), (8, 32, 640, 5120, 1, 5), (8, 32, 2560, 640, 1, 5),
          (8, 16, 1280, 1280, 1, 25), (8, 16, 1280, 3840, 1, 5), (8, 16, 1280, 10240, 1, 5), (8, 16, 5120, 1280, 1, 5),
          (8, 8, 1280, 1280, 1, 5), (8, 8, 1280, 10240, 1, 1), (8, 8, 5120, 1280, 1, 1)]
tot = 0.0
totf = 0.0
for (B, H, Ci, Co, taps, cnt) in shapes:
    M = B * H * H
    dy = torch.randn(M, Co, device="cuda").to(torch.bfloat16)
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    out = torch.zeros(1, Co, taps, Ci, device="cuda")
    geom = (H, H, H, H, 1, 1, 0) if taps == 9 else None
    ms = t(lambda: ops_bwd.gemm_tn(dy, x, out=out, taps=taps, geom=geom, accumulate=True))
    fl = 2.0 * M * Co * Ci * taps
    tot += ms * cnt
    totf += fl * cnt
    print(f"M={M:6d} {Ci:5d}->{Co:5d} taps={taps}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s  x{cnt}", flush=True)
print(f"weighted total {tot:.2f} ms, {totf/tot/1e9:.1f} TF/s")
