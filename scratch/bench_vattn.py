import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, N in [(12, 4096), (4, 4096), (16, 4096), (12, 1024)]:
    qkv = (torch.randn(B, N, 1536, device="cuda")).to(torch.bfloat16)
    q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
    tf = t(lambda: ops.vae_attention(q, k, v))
    qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
    def old():
        s = ops.bmm_nt(qc, kc, out_f32=True)
        p = ops.softmax_rows(s, torch.bfloat16, scale=512 ** -0.5)
        return ops.bmm_nt(p, ops.transpose(vc))
    to = t(old)
    fl = 4.0 * B * N * N * 512
    print(f"B={B} N={N}: flash {tf:8.1f} us {fl/tf/1e6:7.1f} TF/s | materialised {to:8.1f} us {fl/to/1e6:7.1f} TF/s", flush=True)
