import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
torch.manual_seed(0)
def ref(q, k, v):
    s = torch.bmm(q.double(), k.double().transpose(1, 2)) * 512 ** -0.5
    return torch.bmm(torch.softmax(s, -1), v.double()).float()
def rel(a, b): return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
dt = torch.bfloat16
for name, N, mk in [("uniformP N=32", 32, "q0"), ("uniformP N=4096", 4096, "q0"), ("rand N=32", 32, "r"), ("rand N=64", 64, "r"), ("rand N=128", 128, "r"),
                    ("rand N=4096", 4096, "r"), ("onehotV N=32", 32, "oh")]:
    q = torch.randn(1, N, 512) * (0 if mk == "q0" else 1.0)
    k = torch.randn(1, N, 512)
    v = torch.randn(1, N, 512)
    if mk == "oh":
        v = torch.zeros(1, N, 512); v[0, torch.arange(N), torch.arange(N)] = 1.0     # out[i][j] = P[i][j] for j < N
    q, k, v = q.to(dt), k.to(dt), v.to(dt)
    qs = (q.float() * ops.VATTN_QSCALE).to(dt)
    y = ops.vae_attention(qs.cuda(), k.cuda(), v.cuda())
    r = ref(qs.float() / ops.VATTN_QSCALE, k.float(), v.float())
    print(name, "rel", rel(y, r), "finite", bool(torch.isfinite(y).all()))
    if mk == "oh":
        P = y[0, :4, :N].float().cpu(); R = r[0, :4, :N]
        print(" P row0 got", [round(float(x), 3) for x in P[0]]); print(" P row0 ref", [round(float(x), 3) for x in R[0]])
        # find permutation
        for i in range(2):
            order_got = torch.argsort(P[i], descending=True)[:6].tolist(); order_ref = torch.argsort(R[i], descending=True)[:6].tolist()
            print(" top keys got", order_got, "ref", order_ref)
