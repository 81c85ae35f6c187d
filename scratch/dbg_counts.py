import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffews_amd import episodes
pipe, _ = bench.build_pipeline(torch.bfloat16)
bt = episodes.make_episode_batch(4, 1, 512, seed=100, device="cuda")
for cap in (False, True, True):
    r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"], captured=cap)
    d = r["dec"]
    print("captured", cap, "counts", r["counts"].tolist(), "dec min/max", float(d.min()), float(d.max()),
          "u8 max per img", r["seg_u8"].flatten(1).max(1).values.tolist(), "u8 mean", float(r["seg_u8"].float().mean()))
# old-style glue path for comparison
allimg = torch.cat([bt["support_imgs"], bt["support_masks"], bt["query_img"]], 0)
z = pipe.encode_rgb(allimg)
print("z_all vs new path:", float((z[8:] - pipe.encode_rgb(bt["query_img"])).abs().max()))
