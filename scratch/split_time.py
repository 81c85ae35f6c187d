import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffews_amd import episodes
pipe, _ = bench.build_pipeline(torch.bfloat16)
bt = episodes.make_episode_batch(4, 1, 512, seed=1, device="cuda")
def T(f, n=3):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
allimg = torch.cat([bt["support_imgs"], bt["support_masks"], bt["query_img"]], 0)
t_enc, z = T(lambda: pipe.encode_rgb(allimg))
zr, zg, zq = z[:4], z[4:8], z[8:]
cond = torch.cat([zr, zg], 1)
ehs = pipe.empty_text_embed.repeat(4, 1, 1)
t_unet, z0 = T(lambda: pipe.unet.forward_pair(cond, zq.contiguous(), 1, ehs, ehs, out_scale=-1.0))
t_dec, dec = T(lambda: pipe.decode_seg(z0))
print(f"eager: VAE-enc(12 img) {t_enc:.2f} ms | UNet pair {t_unet:.2f} ms | VAE-dec(4 img) {t_dec:.2f} ms | sum {t_enc+t_unet+t_dec:.2f}")
