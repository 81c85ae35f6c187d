"""gemm_big configuration vs gemm_kernel on the shapes gemm_big takes today: one child per setting."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffews_amd import ops
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _cfg
    _cfg.apply_env_config()
    tag = os.environ.get("TAG", "")
    def t(fn):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3
    lins = [(2048, 10240, 1280, True), (32768, 2560, 320, True), (8192, 5120, 640, True), (8192, 1920, 640, False), (2048, 3840, 1280, False),
            (49152, 512, 512, False), (786432, 256, 128, False), (262144, 256, 512, False), (196608, 512, 256, False), (1048576, 128, 256, False),
            (32768, 960, 320, False), (32768, 320, 1280, False), (8192, 640, 2560, False)]
    for (M, N, K, geglu) in lins:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        try:
            us = t(lambda: ops.linear(x, w, bias=b, geglu=geglu) if geglu else ops.linear(x, w, bias=b))
        except Exception as e:
            print(f"lin {tag} {M} {N} {K} ERR {str(e)[:40]}"); continue
        print(f"lin {tag} {M} {N} {K} {us:.1f}", flush=True)
    convs = [(4, 64, 512, 512), (12, 64, 512, 512), (4, 128, 512, 512), (4, 64, 512, 256), (8, 64, 320, 320), (8, 32, 640, 640)]
    for (B, H, Ci, Co) in convs:
        x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16); w = (torch.randn(Co, 9 * Ci, device="cuda") * 0.02).to(torch.bfloat16)
        try:
            us = t(lambda: ops.conv3x3(x, w, Co))
        except Exception as e:
            print(f"conv {tag} {B*H*H} {Co} {9*Ci} ERR {str(e)[:40]}"); continue
        print(f"conv {tag} {B*H*H} {Co} {9*Ci} {us:.1f}", flush=True)
else:
    for tag, env in [("default", {}), ("nobig", {"DFW_CFG": "big_kernels=0"}), ("256x256x32", {"DFW_CFG": "big_bm=256,big_bn=256,big_bk=32"}),
                     ("512x128x32", {"DFW_CFG": "big_bm=512,big_bn=128,big_bk=32"}), ("256x128x64", {"DFW_CFG": "big_bm=256,big_bn=128,big_bk=64"}),
                     ("256x128x32", {"DFW_CFG": "big_bm=256,big_bn=128,big_bk=32"})]:
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TAG=tag, **env))
