"""Training step at BASELINE configs[4] scale on ONE MI355X: SD-2.1 UNet (865.9 M parameters), 512x512 (64x64
latents), nshot-shot episode, batch 1: lock-step forward over [nshot support ; 1 query] latents, MSE, backward, clip +
AdamW.  Synthetic weights / latents / 77-token prompt.  Prints ms per phase (HIP events on the launch stream).
    python scratch/bench_train.py [nshot=7] [steps=3] [dtype=bf16]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffews_amd import config, weights                      # noqa: E402
from diffews_amd.train import UNetTrainer, poly_lr           # noqa: E402

nshot = int(sys.argv[1]) if len(sys.argv) > 1 else 7
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dt = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "fp16") else torch.bfloat16
ucfg = config.get("sd21_unet")
t0 = time.time()
tr = UNetTrainer(ucfg, weights.synthetic_unet_state_dict(ucfg), torch_dtype=dt, loss_scale=1.0 if dt == torch.bfloat16 else 1024.0)
print(f"[train] parameters {tr.P.numel / 1e6:.1f} M (packed, padded), build {time.time() - t0:.1f}s", flush=True)
g = torch.Generator().manual_seed(0)
zr = (torch.randn(nshot, 8, 64, 64, generator=g) * 0.5).cuda()
zq = (torch.randn(1, 4, 64, 64, generator=g) * 0.5).cuda()
tgt = (torch.randn(1, 4, 64, 64, generator=g) * 0.5).cuda()
ehs = torch.randn(1, 77, ucfg["cross_attention_dim"], generator=g).cuda()
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(steps + 2):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    loss, pred = tr.forward_backward(zr, zq, tgt, 1, ehs)
    e1.record()
    tr.optimizer_step(poly_lr(1e-5, it, 1000), max_grad_norm=1.0)
    e2.record()
    torch.cuda.synchronize()
    print(f"[train] step {it}: loss {float(loss):.5f}  fwd+bwd {e0.elapsed_time(e1):8.2f} ms  clip+AdamW+shadow {e1.elapsed_time(e2):7.2f} ms"
          f"  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
