"""The UNet's 3x3 convs below the 64x64 level (lock-step batch of B latents): auto plan against forced tile / split-K.
python scratch/bench_smallconv.py [B] [quick]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
quick = len(sys.argv) > 2
shapes = [(16, 1280, 1280), (16, 2560, 1280), (16, 1920, 1280), (16, 640, 1280), (8, 1280, 1280), (8, 2560, 1280),
          (32, 640, 640), (32, 1280, 640), (32, 1920, 640), (32, 960, 640), (32, 320, 640), (32, 1280, 1280)]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (H, Ci, Co) in shapes:
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    fl = 2.0 * B * H * H * Co * 9 * Ci
    _lib.configure()
    ta = t(lambda: ops.conv3x3(x, w, Co, bias=b))
    line = f"{H:2d}x{H:<2d} {Ci:4d}->{Co:4d} M={B*H*H:5d} K={9*Ci:5d}: auto {ta:7.1f} us {fl/ta/1e6:6.1f} TF/s |"
    if not quick:
        best = (1e9, None)
        for (bm, bn) in [(128, 128), (128, 64), (64, 64)]:
            for sk in (1, 2, 4, 8, 16):
                _lib.configure(big_kernels=0, conv_patch=0, gemm_bm=bm, gemm_bn=bn)
                try:
                    tt = t(lambda: ops.conv3x3(x, w, Co, bias=b, splitk=sk), 10)
                except Exception as e:
                    continue
                if tt < best[0]: best = (tt, (bm, bn, sk))
        line += f" best forced {best[0]:7.1f} us {best[1]}"
    print(line, flush=True)
_lib.configure()
