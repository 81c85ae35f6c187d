"""8-phase GEMM prototype (scratch/proto/gemm8p.hip) vs gemm_big (ops.linear) vs hipBLASLt (torch.matmul), random bf16 data,
interleaved rounds in one process (guide rule 24).  Build: hipcc here (before the GPU is touched)."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgemm8p.so")
SRC = os.path.join(HERE, "gemm8p.hip")
if not os.path.isfile(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC])
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from diffews_amd import ops  # noqa: E402

lib = ctypes.CDLL(SO)
lib.proto_gemm8p.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
VARS = [int(v) for v in os.environ.get("VARS", "0,1,2,4,8,128,130").split(",")]
ABL = [int(v) for v in os.environ.get("ABL", "16,32,64,48,80,96,112").split(",") if v]
SHAPES = [(4096, 4096, 4096), (8192, 8192, 8192), (32768, 512, 4608), (65536, 256, 2304), (16384, 1280, 1280), (8192, 8192, 512)]


def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in SHAPES:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.matmul(a.float()[:512], w.float().t())
    fns = {"gemm_big": lambda: ops.linear(a, w), "hipblaslt": lambda: torch.matmul(a, w.t())}
    for v in VARS:
        c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        rc = lib.proto_gemm8p(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, v, st)
        torch.cuda.synchronize()
        if rc != 0:
            print(f"{M}x{N}x{K} var {v}: rc={rc} (shape not supported)")
            continue
        err = float((c[:512].float() - ref).norm() / ref.norm())
        full = ops.linear(a, w)
        err_full = float((c.float() - full.float()).norm() / full.float().norm())
        print(f"{M}x{N}x{K} var {v}: rel err first rows {err:.2e}, whole vs gemm_big {err_full:.2e}", flush=True)
        fns[f"8p/{v}"] = (lambda v=v, c=c: lib.proto_gemm8p(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, v, st))
    if (M, N, K) in ((4096, 4096, 4096), (32768, 512, 4608)):
        cx = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        for v in ABL:      # ablations: wrong results on purpose, only the time counts
            fns[f"abl/{v}"] = (lambda v=v: lib.proto_gemm8p(a.data_ptr(), w.data_ptr(), cx.data_ptr(), M, N, K, v, st))
    for f in fns.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(5):
        for k, f in fns.items():
            res[k].append(timeit(f))
    fl = 2.0 * M * N * K
    for k, ts in res.items():
        ts = sorted(ts)
        print(f"  {M}x{N}x{K} {k:10s} median {ts[len(ts)//2]:8.1f} us  {fl/ts[len(ts)//2]/1e6:7.1f} TF/s   (min {ts[0]:.1f})", flush=True)
