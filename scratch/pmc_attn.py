"""One shape of the KV-fusion attention for counter passes: python scratch/pmc_attn.py [pre=1]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
pre = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
B, heads, N, nshot = 8, 5, 4096, 1
C = heads * 64
qkv = torch.randn(B, N, 3 * C, device="cuda").to(torch.bfloat16)
bank = torch.randn((B // 2) * nshot, N, 3 * C, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.fsa_attention(qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:], heads, bank[..., C:2*C], bank[..., 2*C:], nshot=nshot,
                      n_plain=B // 2, q_prescaled=pre)
torch.cuda.synchronize()
