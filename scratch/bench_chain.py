"""Does a producer->consumer chain at the VAE's 512^2 level run faster when the batch fits the 256 MB
Infinity Cache?  One resnet (norm1-conv1-norm2-conv2+x) on B images at once vs in chunks."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, packing
def t(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
C, H = 128, 512
w1 = packing.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5).to("cuda", torch.bfloat16)
w2 = packing.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5).to("cuda", torch.bfloat16)
b1, b2 = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
def resnet(x):
    h = ops.groupnorm(x, g, be, 32, 1e-6, silu=True)
    h = ops.conv3x3(h, w1, C, bias=b1, gn_groups=32)
    h = ops.groupnorm(h, g, be, 32, 1e-6, silu=True)
    return ops.conv3x3(h, w2, C, bias=b2, residual=x, gn_groups=32)
def two(x):
    return resnet(resnet(x))
for B in (12, 4):
    x = torch.randn(B, H, H, C, device="cuda", dtype=torch.bfloat16)
    full = t(lambda: two(x))
    for ch in (1, 2, 3, 4, 6):
        if B % ch: continue
        parts = [x[i:i + ch] for i in range(0, B, ch)]
        ms = t(lambda: [two(p) for p in parts])
        print(f"B={B}: chunks of {ch} ({ch * H * H * C * 2 / 1e6:.0f} MB/tensor): {ms:.3f} ms   (all at once {full:.3f} ms)", flush=True)
