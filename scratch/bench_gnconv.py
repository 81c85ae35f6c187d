"""GroupNorm+SiLU+conv3x3 on the VAE resnet shapes; DFW_GN_FUSE=1 for the fused kernel, default = groupnorm pass + conv."""
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

tot = 0
for (B, H, Cin, Cout, cnt) in [(12, 512, 128, 128, 4), (12, 256, 128, 256, 1), (12, 256, 256, 256, 3),
                               (4, 512, 256, 128, 1), (4, 512, 128, 128, 5), (4, 256, 256, 256, 5)]:
    x = torch.randn(B, H, H, Cin, device="cuda", dtype=torch.bfloat16)
    w = packing.pack_conv3x3(torch.randn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).to("cuda", torch.bfloat16)
    b = torch.randn(Cout, device="cuda")
    g, be = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    res = torch.randn(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: ops.conv3x3(x, w, Cout, bias=b, residual=res, gn_groups=32, gn_in=(g, be, 32, 1e-6, os.environ.get("NOSILU") is None)))
    fl = 2 * B * H * H * Cout * 9 * Cin
    tot += ms * cnt
    print(f"B{B} {H}^2 {Cin}->{Cout}: {ms*1e3:.0f} us  ({fl/ms/1e12:.3f} PFLOP/s incl. norm)  x{cnt}")
print(f"weighted total {tot:.2f} ms")
