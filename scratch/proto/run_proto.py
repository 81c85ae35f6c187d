"""One-wave-per-SIMD GEMM prototype vs the library's gemm_big on the same NT GEMM (random bf16 data)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffews_amd import ops
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libproto.so"))
lib.proto_gemm1w.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(8192, 8192, 4096), (65536, 512, 4608), (196608, 512, 4608), (262144, 256, 2304)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.proto_gemm1w(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, st)
    torch.cuda.synchronize()
    ref = ops.linear(a, w)
    err = float((c.float() - ref.float()).norm() / ref.float().norm())
    tp = t(lambda: lib.proto_gemm1w(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, st))
    tb = t(lambda: ops.linear(a, w))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: rc={rc} err={err:.1e} proto {tp*1e3:8.1f} us {fl/tp/1e9:7.1f} TF/s | gemm_big {tb*1e3:8.1f} us {fl/tb/1e9:7.1f} TF/s", flush=True)
