"""Does the 256 MB Infinity Cache help a producer -> GroupNorm -> consumer chain when the VAE runs in image groups?
conv(128->128) -> groupnorm(fused stats) -> conv chain at 512^2, total 12 images processed in groups of g."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, packing

def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

C = 128
w = packing.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5).to("cuda", torch.bfloat16)
b = torch.randn(C, device="cuda")
g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
X = torch.randn(12, 512, 512, C, device="cuda", dtype=torch.bfloat16)
for grp in (12, 6, 4, 3, 2, 1):
    def run():
        for i in range(0, 12, grp):
            x = X[i:i + grp]
            h = x
            for _ in range(4):       # 2 resnets: (gn, conv) x 4
                h = ops.conv3x3(h, w, C, bias=b, gn_groups=32, gn_in=(g, be, 32, 1e-6, True))
    ms = t(run)
    print(f"group {grp:2d}: {ms:.3f} ms for 12 images x 4 (gn+conv)", flush=True)
