"""Loss-error statistic of the sampled-VAE training parity case under k8 = 0 (round-3 kernels) and k8 = 3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from diffews_amd import _lib
import test_fullsize_gpu as T
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
for k8 in (0, 3):
    _lib.configure(k8=k8)
    for seed in (62, 63):
        r = T._fullsize_train_case(torch.bfloat16, 2, 512, True, seed=seed)
        print(f"k8={k8} seed={seed}: latents {r['latents']:.3e} pred {r['pred']:.3e} loss {r['loss']:.3e} flat {r['flat_rel']:.3e}", flush=True)
