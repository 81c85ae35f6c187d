"""gemm_big's minimum tile count: the UNet's 3x3 convs and linears whose 256-wide tilings give 64..191 tiles, per threshold.
python scratch/bench_mintiles.py B"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
convs = [(32, 640, 640), (32, 1280, 640), (32, 1920, 640), (32, 960, 640), (32, 320, 640), (16, 1280, 1280), (16, 2560, 1280),
         (16, 1920, 1280), (16, 640, 1280), (64, 320, 320), (64, 640, 320), (64, 960, 320)]
lins = [(64 * 64, 320, 320), (64 * 64, 320, 960), (32 * 32, 640, 640), (32 * 32, 640, 1920), (32 * 32, 640, 5120), (32 * 32, 2560, 640),
        (16 * 16, 1280, 1280), (16 * 16, 1280, 3840), (16 * 16, 1280, 10240), (16 * 16, 5120, 1280)]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ths = (192, 160, 128, 96, 64)
for (H, Ci, Co) in convs:
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    fl = 2.0 * B * H * H * Co * 9 * Ci
    line = f"conv {H:2d}x{H:<2d} {Ci:4d}->{Co:4d} M={B*H*H:6d}:"
    for th in ths:
        _lib.configure(big_min_tiles=th)
        tt = t(lambda: ops.conv3x3(x, w, Co, bias=b))
        line += f"  >={th}: {tt:7.1f} us {fl/tt/1e6:6.0f}TF"
    print(line, flush=True)
for (n, K, N) in lins:
    x = torch.randn(B * n, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    fl = 2.0 * B * n * N * K
    line = f"lin  M={B*n:6d} K={K:5d} N={N:5d}:"
    for th in ths:
        _lib.configure(big_min_tiles=th)
        tt = t(lambda: ops.linear(x, w, bias=b))
        line += f"  >={th}: {tt:7.1f} us {fl/tt/1e6:6.0f}TF"
    print(line, flush=True)
_lib.configure()
