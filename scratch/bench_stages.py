"""gemm_kernel LDS ring depth (dfw_config.gemm_stages) x tile on the UNet's token GEMMs and small convs, timed as a replayed
HIP graph of 20 dependent launches (what the captured step sees: launch gaps included), interleaved rounds (guide rule 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffews_amd import _lib, ops

dt = torch.bfloat16
dev = "cuda"
# (M, N, K, epilogue) of the lock-step batch of 8 latents (bench default, 512^2, 1-shot, batch 4)
LIN = [(32768, 320, 320, "b"), (32768, 320, 320, "br"), (8192, 640, 640, "br"), (2048, 1280, 1280, "br"), (512, 1280, 1280, "br"),
       (32768, 320, 1280, "br"), (8192, 640, 2560, "br"), (2048, 1280, 5120, "br"), (512, 1280, 5120, "br"),
       (32768, 960, 320, ""), (8192, 1920, 640, ""), (2048, 3840, 1280, ""), (512, 3840, 1280, ""),
       (32768, 2560, 320, "g"), (8192, 5120, 640, "g"), (2048, 10240, 1280, "g")]
CONV = [(8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280), (8, 16, 2560, 1280), (8, 32, 1280, 640)]
ONLY = os.environ.get("ONLY", "lin,conv").split(",")
N_CHAIN = 20


def graph_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N_CHAIN):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / N_CHAIN * 1e3)
    return sorted(ts)[len(ts) // 2]


def variants():
    v = [("default", {})]
    for (bm, bn) in ((64, 64), (128, 64), (128, 128)):
        for s in (2, 3, 4):
            if s == 4 and bm + bn > 192:
                continue
            v.append((f"{bm}x{bn}/S{s}", dict(big_kernels=0, conv_patch=0, gemm_bm=bm, gemm_bn=bn, gemm_stages=s)))
    return v


def run(label, fn, flops):
    res = {}
    for name, cfg in variants():
        _lib.configure(); _lib.configure(**cfg) if cfg else None
        try:
            res[name] = graph_time(fn)
        except RuntimeError as e:
            res[name] = None
    _lib.configure()
    best = min((t, n) for n, t in res.items() if t)
    print(f"{label:42s} " + "  ".join(f"{n} {t:6.1f}" if t else f"{n}   --  " for n, t in res.items())
          + f"   | best {best[1]} {best[0]:.1f} us = {flops / best[0] / 1e6:.0f} TF/s", flush=True)


if "lin" in ONLY:
    for (M, N, K, epi) in LIN:
        x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        ge = "g" in epi
        b = torch.randn(N, device=dev) if ("b" in epi or ge) else None
        r = (torch.randn(M, N, device=dev) * 0.5).to(dt) if "r" in epi else None
        out = torch.empty(M, N // 2 if ge else N, device=dev, dtype=dt)
        run(f"lin {M}x{N}x{K} [{epi}]", lambda: ops.linear(x, w, bias=b, residual=r, geglu=ge, out=out), 2.0 * M * N * K)
if "conv" in ONLY:
    from diffews_amd import packing
    for (B, H, Cin, Cout) in CONV:
        x = (torch.randn(B, H, H, Cin, device=dev) * 0.5).to(dt)
        w = packing.pack_conv3x3((torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)).to(dt).contiguous()
        b = torch.randn(Cout, device=dev)
        run(f"conv {B}x{H}x{H} {Cin}->{Cout}", lambda: ops.conv3x3(x, w, Cout, bias=b), 2.0 * B * H * H * Cout * Cin * 9)
