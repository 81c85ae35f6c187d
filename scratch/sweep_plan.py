"""(tile, split-K) sweep of gemm_kernel on the UNet's shapes at the lock-step batch of 8 latents, for the plan_gemm fit.
python scratch/sweep_plan.py > gpurun_out/sweep_plan.txt   (spawns one child per forced tile)"""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffews_amd import ops
    tile = os.environ["SWEEP_TILE"]
    from diffews_amd import _lib
    _lib.configure(big_kernels=0, gemm_bm=int(tile.split("x")[0]), gemm_bn=int(tile.split("x")[1]))
    convs = [(8, 64, 320, 320), (8, 64, 640, 320), (8, 64, 960, 320), (8, 32, 640, 640), (8, 32, 320, 640), (8, 32, 1280, 640),
             (8, 32, 1920, 640), (8, 32, 960, 640), (8, 16, 1280, 1280), (8, 16, 640, 1280), (8, 16, 2560, 1280), (8, 16, 1920, 1280),
             (8, 8, 1280, 1280), (8, 8, 2560, 1280)]
    lins = [(32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (32768, 960, 320), (8192, 1920, 640), (2048, 3840, 1280),
            (32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120), (512, 1280, 1280), (32768, 64, 320), (8192, 64, 640),
            (32768, 320, 64), (2048, 64, 1280), (512, 1280, 5120)]
    def t(fn):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3
    for (B, H, Ci, Co) in convs:
        x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16); w = (torch.randn(Co, 9 * Ci, device="cuda") * 0.02).to(torch.bfloat16)
        for sk in (1, 2, 4, 8, 16):
            if 9 * Ci // 32 // sk < 8: continue
            try:
                us = t(lambda: ops.conv3x3(x, w, Co, splitk=sk))
            except Exception as e:
                continue
            print(f"conv {tile} {B*H*H} {Co} {9*Ci} {sk} {us:.1f}", flush=True)
    for (M, N, K) in lins:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        for sk in (1, 2, 4, 8, 16):
            if K // 32 // sk < 8 and sk > 1: continue
            try:
                us = t(lambda: ops.linear(x, w, splitk=sk))
            except Exception as e:
                continue
            print(f"lin {tile} {M} {N} {K} {sk} {us:.1f}", flush=True)
else:
    for tile in ("128x128", "128x64", "64x64"):
        env = dict(os.environ, SWEEP_TILE=tile)
        subprocess.run([sys.executable, __file__, "child"], env=env)
