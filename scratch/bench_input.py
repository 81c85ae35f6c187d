"""Input-transform throughput: GPU EpisodeLoader vs the reference's host path (PIL resize + ToTensor +
Normalize + nearest mask), on already-decoded 640x480 RGB arrays.  python scratch/bench_input.py"""
import sys, os, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from diffews_amd.input_pipeline import EpisodeLoader

S, b, s, n = 512, 4, 1, 200
rng = np.random.default_rng(0)
pool = [(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), rng.integers(0, 3, (480, 640)).astype(np.uint8)) for _ in range(16)]
def eps(n):
    for i in range(n):
        q, sp = pool[i % 16], pool[(i + 5) % 16]
        yield dict(query_img=q[0], query_mask=q[1], support_imgs=[sp[0]] * s, support_masks=[sp[1]] * s, class_id=0)

for _ in EpisodeLoader(eps(8), S, b, s): pass
torch.cuda.synchronize(); t0 = time.time()
cnt = 0
for batch in EpisodeLoader(eps(n), S, b, s):
    cnt += batch["query_img"].shape[0]
torch.cuda.synchronize(); dt = time.time() - t0
print(f"GPU loader: {cnt/dt:.1f} episodes/s ({cnt*(s+1)/dt:.0f} images/s + masks), 1 host thread packing")

def host(ep):
    def im(x):
        r = np.asarray(Image.fromarray(x, "RGB").resize((S, S), Image.BILINEAR))
        t = torch.from_numpy(r.copy()).permute(2, 0, 1).float().div(255)
        return (t - 0.5) / 0.5
    def mk(m):
        return F.interpolate(torch.from_numpy((m == 1).astype(np.float32))[None, None], (S, S), mode="nearest")[0, 0]
    return im(ep["query_img"]), mk(ep["query_mask"]), [im(x) for x in ep["support_imgs"]], [mk(x) for x in ep["support_masks"]]
torch.set_num_threads(1)
t0 = time.time(); c = 0
for e in eps(40):
    host(e); c += 1
dt = time.time() - t0
print(f"host PIL/torch path: {c/dt:.1f} episodes/s on 1 thread")
