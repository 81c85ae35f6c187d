"""One shape of the KV-fusion attention backward for counter passes: python scratch/pmc_attn_bwd.py [N heads nshot]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops, ops_bwd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nshot = int(sys.argv[3]) if len(sys.argv) > 3 else 7
B, n_plain = nshot + 1, nshot
C = heads * 64
qkv = (torch.randn(B, N, 3 * C, device="cuda") * 0.5).to(torch.bfloat16)
dout = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
out = ops.fsa_attention(q, k, v, heads, k[:n_plain], v[:n_plain], nshot=nshot, n_plain=n_plain, q_prescaled=True, lse=lse)
for _ in range(3):
    ops_bwd.fsa_attention_bwd(qkv, out, dout, lse, heads, nshot=nshot, n_plain=n_plain)
torch.cuda.synchronize()
