import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _cfg
_cfg.apply_env_config()
shapes = {"vae512": (12, 512, 512, 128, 128), "vae256": (12, 256, 256, 256, 256), "vae128": (12, 128, 128, 512, 512),
          "unet64": (4, 64, 64, 320, 320), "unet32": (4, 32, 32, 640, 640), "unet16": (4, 16, 16, 1280, 1280),
          "unet8": (4, 8, 8, 1280, 1280), "dec512": (4, 512, 512, 128, 128), "b1_512": (1, 512, 512, 128, 128), "b2_512": (2, 512, 512, 128, 128),
          "b24_512": (24, 512, 512, 128, 128), "k2304n128": (12, 256, 256, 256, 128), "k1152n256": (12, 256, 256, 128, 256),
          "dec64": (4, 64, 64, 512, 512), "enc64": (12, 64, 64, 512, 512), "dec128": (4, 128, 128, 512, 512), "unet32": (8, 32, 32, 640, 640),
          "unet64b8": (8, 64, 64, 320, 320), "unet64k2": (8, 64, 64, 640, 320), "unet64k3": (8, 64, 64, 960, 320), "dec256": (4, 256, 256, 256, 256)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(shapes)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for n in names:
    B, H, W, Ci, Co = shapes[n]
    x = torch.randn(B, H, W, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    for _ in range(2):
        y = ops.conv3x3(x, w, Co, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = ops.conv3x3(x, w, Co, bias=b)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    fl = 2.0 * B * H * W * Co * 9 * Ci
    print(f"{n:8s} M={B*H*W:8d} N={Co:5d} K={9*Ci:6d}  {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s", flush=True)
