import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffews_amd import episodes
from diffews_amd.metrics import AverageMeter, fold_class_ids
pipe, _ = bench.build_pipeline(torch.bfloat16)
b, s, res = 4, 1, 512
bt = episodes.make_episode_batch(b, s, res, seed=100, device="cuda")
cls = episodes.episode_class_ids(list(range(b))).cuda()
meter = AverageMeter("coco", fold_class_ids("coco", 0), device="cuda")
def step(captured):
    r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"], captured=captured)
    meter.update_from_counts(r["counts"], cls)
    return r
step(False); torch.cuda.synchronize()
print("after eager:", meter.intersection_buf[:, [0,4,8,12]].tolist(), meter.union_buf[:, [0,4,8,12]].tolist())
r = step(True)
print("after captured 1:", meter.intersection_buf[:, [0,4,8,12]].tolist(), meter.union_buf[:, [0,4,8,12]].tolist(), r["counts"].tolist())
stat = pipe.episode_input_buffers(b, s, res)
print("stat keys", list(stat.keys()), "gt equal", torch.equal(stat["query_gt"], bt["query_mask"]), stat["query_gt"].dtype)
bt = dict(support_imgs=stat["support_imgs"], query_img=stat["query_img"], support_masks=stat["support_masks"], query_mask=stat["query_gt"])
meter.intersection_buf.zero_(); meter.union_buf.zero_()
for _ in range(3):
    r = step(True)
torch.cuda.synchronize()
print("after 3 static:", meter.intersection_buf[:, [0,4,8,12]].tolist(), meter.union_buf[:, [0,4,8,12]].tolist(), r["counts"].tolist())
print(meter.compute_iou()[:2])
