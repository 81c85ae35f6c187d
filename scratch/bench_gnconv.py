"""conv3x3(gn_in=...) fused in conv_patch8_kernel<.,.,GNIN> vs GroupNorm pass + conv (round 4).  us per call, torch events.
Needs scratch/conv_patch8_gnin_fused.patch.txt applied (ops.groupnorm_coef / ops.fuse_gn_in are not in the product)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _cfg
_cfg.apply_env_config()
shapes = {"vae256": (12, 256, 256, 256, 256), "vae128": (12, 128, 128, 512, 512), "dec256": (4, 256, 256, 256, 256),
          "k1152n256": (12, 256, 256, 128, 256), "dec64": (4, 64, 64, 512, 512), "dec128": (4, 128, 128, 512, 512)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(shapes)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n in names:
    B, H, W, Ci, Co = shapes[n]
    x = (torch.randn(B, H, W, Ci, device="cuda") * 2 + 0.3).to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    g, be = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    gi = (g, be, 32, 1e-6, True)
    xn = ops.groupnorm(x, g, be, 32, 1e-6, silu=True)
    t_conv = timed(lambda: ops.conv3x3(xn, w, Co, bias=b))
    t_gn = timed(lambda: ops.groupnorm(x, g, be, 32, 1e-6, silu=True))
    t_coef = timed(lambda: ops.groupnorm_coef(x, g, be, 32, 1e-6))
    ops.fuse_gn_in = False
    t_two = timed(lambda: ops.conv3x3(x, w, Co, bias=b, gn_in=gi))
    ops.fuse_gn_in = True
    t_fused = timed(lambda: ops.conv3x3(x, w, Co, bias=b, gn_in=gi))
    print(f"{n:10s} conv {t_conv:7.1f}  groupnorm {t_gn:7.1f}  coef-only {t_coef:7.1f} | two-pass {t_two:7.1f}  fused {t_fused:7.1f} us"
          f"  (fused conv alone ~ {t_fused - t_coef:7.1f})", flush=True)
