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
for (B, H, Cin, Cout) in [(12, 512, 128, 128), (4, 512, 128, 128), (4, 512, 256, 128)]:
    x = torch.randn(B, H, H, Cin, device="cuda", dtype=torch.bfloat16)
    w = packing.pack_conv3x3(torch.randn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).to("cuda", torch.bfloat16)
    b = torch.randn(Cout, device="cuda")
    res = torch.randn(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    fl = 2 * B * H * H * Cout * 9 * Cin
    ms = t(lambda: ops.conv3x3(x, w, Cout, bias=b, residual=res, gn_groups=32))
    print(f"{os.environ.get('DFW_BIG_CFG','default'):14s} B{B} {H}^2 {Cin}->{Cout}: {ms*1e3:7.0f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
