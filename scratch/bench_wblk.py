"""conv_patch with the weight read from the row-major [N][9 Cin] layout vs the blocked [9][Cin/32][N][32] copy
(whole-line W DMA), interleaved rounds in one process, random bf16 data."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, packing
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (B, H, Cin, Cout): VAE 512^2 x 12 128->128, 256^2 x 12 256->256, 128^2 x 12 512->512, UNet 64^2 x 8 320->320 (not /256), decoder 4 img
shapes = [(12, 512, 128, 128), (12, 256, 256, 256), (12, 128, 512, 512), (4, 256, 512, 512), (4, 512, 256, 256), (4, 512, 128, 128), (12, 256, 128, 256), (4, 512, 256, 128)]
for (B, H, Ci, Co) in shapes:
    x = (torch.rand(B, H, H, Ci, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(Co, 9 * Ci, device="cuda") * 2 - 1) * (9 * Ci) ** -0.5).to(torch.bfloat16)
    wb = packing.block_conv3x3(w)
    b = torch.randn(Co, device="cuda")
    y0 = ops.conv3x3(x, w, Co, bias=b)
    y1 = ops.conv3x3(x, w, Co, bias=b, w_blk=wb)
    assert torch.equal(y0, y1), (B, H, Ci, Co, float((y0.float() - y1.float()).abs().max()))
    f = {"rowmajor": lambda: ops.conv3x3(x, w, Co, bias=b), "blocked": lambda: ops.conv3x3(x, w, Co, bias=b, w_blk=wb)}
    for fn in f.values():
        for _ in range(3): fn()
    r = {k: [] for k in f}
    for _ in range(5):
        for k, fn in f.items(): r[k].append(t(fn))
    fl = 2.0 * B * H * H * Co * 9 * Ci
    s = "  ".join(f"{k} {sorted(v)[2]:8.1f} us {fl / sorted(v)[2] / 1e6:7.1f} TF/s" for k, v in r.items())
    print(f"B={B} {H}x{H} {Ci}->{Co}: {s}   ratio {sorted(r['rowmajor'])[2] / sorted(r['blocked'])[2]:.3f}", flush=True)
