"""random vs zero-filled operands on the two dominant conv shapes (clock / power response vs structure)."""
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
for (B, H, Cin, Cout) in [(12, 512, 128, 128), (12, 256, 256, 256), (12, 128, 512, 512)]:
    fl = 2 * B * H * H * Cout * 9 * Cin
    for name in ("random", "zeros"):
        x = torch.randn(B, H, H, Cin, device="cuda", dtype=torch.bfloat16)
        w = packing.pack_conv3x3(torch.randn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).to("cuda", torch.bfloat16)
        if name == "zeros":
            x.zero_(); w.zero_()
        ms = t(lambda: ops.conv3x3(x, w, Cout))
        print(f"B{B} {H}^2 {Cin}->{Cout} {name:7s}: {ms*1e3:7.0f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
