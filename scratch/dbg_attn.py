import torch, torch.nn.functional as F, sys
sys.path.insert(0, '.')
from diffews_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
B, N, C = 1, 64, 64
q = torch.randn(B, N, C).to(dt); k = torch.randn(B, N, C).to(dt); v = torch.randn(B, N, C).to(dt)
def ref(q,k,v): return F.scaled_dot_product_attention(q.float()[:,None], k.float()[:,None], v.float()[:,None])[:,0]
def run(q,k,v): return ops.fsa_attention(q.cuda(), k.cuda(), v.cuda(), 1).float().cpu()
# 1) V = ones -> out must be 1
y = run(q, k, torch.ones_like(v)); print("V=1: min/max", y.min().item(), y.max().item())
# 2) K = 0 -> uniform -> out = mean(V)
y = run(q, torch.zeros_like(k), v); print("K=0 err", (y - v.float().mean(1, keepdim=True)).abs().max().item())
# 3) V = one-hot key index in channel (key j -> e_j): out[q][j] = P[q][j]
eye = torch.eye(64)[None].to(dt)
y = run(q, k, eye); P = torch.softmax(q.float() @ k.float().transpose(1,2) * 0.125, -1)
print("P err", (y - P).abs().max().item(), "rowsum", y.sum(-1)[0,:8])
d = (y-P).abs()[0]
print("bad cols per row0:", (d[0] > 1e-2).nonzero().flatten().tolist())
print("y row0", y[0,0,:16]); print("P row0", P[0,0,:16])
