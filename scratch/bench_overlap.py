"""Does a HBM-bound gn_apply hide beside an MFMA-bound conv of ANOTHER half batch on a second stream?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, packing
C = 128
w = packing.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5).to("cuda", torch.bfloat16)
b = torch.randn(C, device="cuda")
g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
xa = torch.randn(6, 512, 512, C, device="cuda", dtype=torch.bfloat16)
xb = torch.randn(6, 512, 512, C, device="cuda", dtype=torch.bfloat16)
ya = ops.conv3x3(xa, w, C, bias=b, gn_groups=32)     # carries fused stats
yb = ops.conv3x3(xb, w, C, bias=b, gn_groups=32)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def conv(x): return ops.conv3x3(x, w, C, bias=b, gn_groups=32)
def gn(y): return ops.groupnorm(y, g, be, 32, 1e-6, silu=True)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def serial():
    conv(xa); gn(yb)
def overlapped():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): conv(xa)
    with torch.cuda.stream(s2): gn(yb)
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"conv alone {timeit(lambda: conv(xa)):.0f} us, gn alone {timeit(lambda: gn(yb)):.0f} us")
print(f"serial {timeit(serial):.0f} us, two streams {timeit(overlapped):.0f} us")
# a chain: 4 x (gn + conv) per half, halves interleaved on two streams vs one stream
def chain(x, n=4):
    h = x
    for _ in range(n):
        h = ops.conv3x3(h, w, C, bias=b, gn_groups=32, gn_in=(g, be, 32, 1e-6, True))
    return h
def chain_serial():
    chain(ya); chain(yb)
def chain_2s():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): chain(ya)
    with torch.cuda.stream(s2): chain(yb)
    cur.wait_stream(s1); cur.wait_stream(s2)
x12 = torch.cat([ya, yb])
x12._gn_stats = None
y12 = ops.conv3x3(torch.cat([xa, xb]), w, C, bias=b, gn_groups=32)
print(f"chain 12 images one batch {timeit(lambda: chain(y12)):.0f} us; two halves serial {timeit(chain_serial):.0f} us; two halves on two streams {timeit(chain_2s):.0f} us")
