"""Correctness of the patch conv3x3 (DFW_CONV_PATCH=1) against torch fp32 on the same 16-bit inputs, incl. fused GN stats."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, _lib as L
import torch.nn.functional as F
cases = [(1, 512, 512, 128, 128, True, True), (2, 256, 256, 256, 256, True, True), (3, 256, 256, 256, 128, False, True),
         (2, 128, 128, 512, 512, True, False), (4, 64, 64, 320, 384, False, True), (1, 512, 512, 64, 128, True, True),
         (2, 256, 512, 128, 128, True, True), (12, 512, 512, 128, 128, True, True)]
torch.manual_seed(0)
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for (B, H, W, Ci, Co, res, gn) in cases:
        x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
        w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(dt)
        b = torch.randn(Co, device="cuda")
        r = torch.randn(B, H, W, Co, device="cuda").to(dt) if res else None
        y = ops.conv3x3(x, w, Co, bias=b, residual=r, gn_groups=32 if gn else 0)
        w4 = w.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
        nb = min(B, 2)
        ref = F.conv2d(x[:nb].float().permute(0, 3, 1, 2), w4, b, padding=1).permute(0, 2, 3, 1)
        if res: ref = ref + r[:nb].float()
        err = float((y[:nb].float() - ref).abs().max() / ref.abs().max())
        # last image too
        ref2 = F.conv2d(x[-1:].float().permute(0, 3, 1, 2), w4, b, padding=1).permute(0, 2, 3, 1)
        if res: ref2 = ref2 + r[-1:].float()
        err2 = float((y[-1:].float() - ref2).abs().max() / ref2.abs().max())
        st = getattr(y, "_gn_stats", None)
        serr = -1.0
        if st is not None:
            part, chunks, g = st
            s = part.sum(1)                     # [B, g, 2]
            yf = y.float().view(B, H * W, g, Co // g)
            s_ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], -1)
            # kernel sums the fp32 pre-rounding values; compare loosely
            serr = float(((s - s_ref).abs() / (s_ref.abs() + 1.0)).max())
        tol = 6e-3 if dt == torch.bfloat16 else 8e-4
        ok = err < tol and err2 < tol and serr < 2e-2
        bad += not ok
        print(f"{str(dt)[6:]:8s} B{B} {H}x{W} {Ci}->{Co} res={int(res)} err={err:.2e} {err2:.2e} gn={serr:.2e} {'ok' if ok else 'FAIL'}", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
