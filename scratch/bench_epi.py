"""Does the conv epilogue's HBM traffic show? conv3x3 with / without residual, with / without fused stats."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, packing
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, Cin, Cout) in [(12, 512, 128, 128), (4, 512, 128, 128), (12, 256, 256, 256), (12, 64, 512, 512)]:
    x = torch.randn(B, H, H, Cin, device="cuda", dtype=torch.bfloat16)
    w = packing.pack_conv3x3(torch.randn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).to("cuda", torch.bfloat16)
    b = torch.randn(Cout, device="cuda")
    res = torch.randn(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    fl = 2 * B * H * H * Cout * 9 * Cin
    for name, kw in [("plain", {}), ("+res", dict(residual=res)), ("+stats", dict(gn_groups=32)), ("+res+stats", dict(residual=res, gn_groups=32))]:
        ms = t(lambda: ops.conv3x3(x, w, Cout, bias=b, **kw))
        print(f"B{B} {H}^2 {Cin}->{Cout} {name:11s}: {ms*1e3:7.0f} us  {fl/ms/1e9:7.1f} TF/s")
