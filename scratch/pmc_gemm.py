"""One plain GEMM / conv for counter passes: python scratch/pmc_gemm.py lin|conv"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "lin"
if kind == "lin":
    M, N, K = 4096, 4096, 4096
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    for _ in range(5): ops.linear(x, w)
else:
    B, H, Ci, Co = 12, 128, 512, 512
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
    for _ in range(5): ops.conv3x3(x, w, Co)
torch.cuda.synchronize()
