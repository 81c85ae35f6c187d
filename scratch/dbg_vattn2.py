import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
N = 32
q = torch.zeros(1, N, 512).to(dt); k = torch.randn(1, N, 512).to(dt)
v = torch.randn(1, N, 512).to(dt)
y = ops.vae_attention(q.cuda(), k.cuda(), v.cuda()).float().cpu()
r = v.float().mean(1, keepdim=True).expand(1, N, 512)
err = (y - r)[0]          # [N, 512]
print("per d-block (32 cols) rel err:", [round(float(err[:, 32*b:32*b+32].norm() / r[0][:, 32*b:32*b+32].norm()), 3) for b in range(16)])
print("per 8-col chunk err of block 1:", [round(float(err[:, 32+8*c:40+8*c].norm() / r[0][:, 32+8*c:40+8*c].norm()), 3) for c in range(4)])
print("per query row err:", [round(float(err[i].norm() / r[0][i].norm()), 3) for i in range(0, 32, 4)])
# which source column does output column d equal? use V with column-dependent constant: V[j][d] = d
v2 = torch.arange(512).float()[None, None, :].expand(1, N, 512).contiguous().to(dt)
y2 = ops.vae_attention(q.cuda(), k.cuda(), v2.cuda()).float().cpu()[0, 0]
bad = [(d, int(y2[d])) for d in range(512) if abs(float(y2[d]) - d) > 2]
print("col-id test mismatches (out col, value):", bad[:40], len(bad))
# key-dependent constant V[j][d] = j, P one-hot-ish is hard; instead V[j][d] = j with uniform P -> mean 15.5 everywhere
