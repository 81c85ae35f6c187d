"""The UNet's linear shapes (lock-step batch of 8 latents): python scratch/bench_lin.py"""
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
tot = 0
for (M, N, K, cnt, res) in [(32768, 320, 320, 25, True), (8192, 640, 640, 25, True), (2048, 1280, 1280, 25, True),
                            (32768, 960, 320, 5, False), (8192, 1920, 640, 5, False), (2048, 3840, 1280, 5, False),
                            (32768, 320, 1280, 5, True), (8192, 640, 2560, 5, True), (2048, 1280, 5120, 5, True),
                            (512, 1280, 1280, 5, True)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    ms = t(lambda: ops.linear(a, w, bias=b, residual=r))
    tot += ms * cnt
    byts = (M * K + N * K + M * N * (2 if res else 1)) * 2
    print(f"{M:6d}x{N:5d}x{K:5d} x{cnt:2d}: {ms*1e3:7.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s  {byts/ms/1e9:6.2f} TB/s", flush=True)
print(f"weighted total {tot:.3f} ms")
