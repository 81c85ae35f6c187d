import sys, os, subprocess, itertools
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffews_amd import ops
    convs = [(4,16,16,1280,1280), (4,8,8,1280,1280), (4,32,32,640,640), (4,16,16,2560,1280), (4,64,64,320,320)]
    lins = [(1024,1280,1280), (4096,640,640), (16384,320,320), (1024,1280,5120), (256,1280,1280), (4096,640,2560), (16384,960,320)]
    for sk in (1, 2, 4, 8):
        for c in convs:
            B,H,W,Ci,Co = c
            x = torch.randn(B,H,W,Ci,device="cuda").to(torch.bfloat16); w = torch.randn(Co,9*Ci,device="cuda").to(torch.bfloat16)
            try:
                for _ in range(2): ops.conv3x3(x,w,Co,splitk=sk)
            except Exception as e:
                print("conv",c,sk,"ERR",str(e)[:60]); continue
            torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv3x3(x,w,Co,splitk=sk)
            e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/20*1e-3
            print(f"conv M={B*H*W:6d} N={Co:5d} K={9*Ci:6d} sk={sk} {t*1e6:8.1f} us {2.0*B*H*W*Co*9*Ci/t/1e12:7.1f} TF/s", flush=True)
        for (M,N,K) in lins:
            x = torch.randn(M,K,device="cuda").to(torch.bfloat16); w = torch.randn(N,K,device="cuda").to(torch.bfloat16)
            for _ in range(2): ops.linear(x,w,splitk=sk)
            torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.linear(x,w,splitk=sk)
            e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/20*1e-3
            print(f"lin  M={M:6d} N={N:5d} K={K:6d} sk={sk} {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF/s", flush=True)
else:
    for tile in ("128x128", "128x64", "64x64"):
        print("=== tile", tile, flush=True)
        env = dict(os.environ, DFW_GEMM_TILE=tile)
        subprocess.run([sys.executable, __file__, "child"], env=env)
