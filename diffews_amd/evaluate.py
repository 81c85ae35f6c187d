"""Episode-sharded evaluation loop: counterpart of `test_diffusion`
(/root/reference/evaluation_util/main_oss.py:84-171) for one process per GPU.

The reference evaluates on one GPU (E:308) with the threshold and the metric on the host
(E:128-137); here rank r runs episodes i == r (mod R), thresholding + inter/union counts stay on the
device, and the ranks meet once, in a sum all-reduce of the two `[2, nclass]` buffers
(evaluation_util/common/logger.py:30-31).  No data-path collective.
"""
import torch

from . import episodes as ep
from .metrics import AverageMeter, fold_class_ids


def episode_batches(indices, batch):
    for i in range(0, len(indices), batch):
        yield indices[i:i + batch]


@torch.no_grad()
def test_diffusion(pipe, n_episodes, nshot=1, res=512, batch=1, benchmark="coco", fold=0, r_threshold=0.25,
                   rank=0, world_size=1, make_batch=None, device=None, episodes=None, threshold=0.0,
                   batch_max=False, captured=True):
    """Run `n_episodes` episodes sharded over `world_size` ranks; returns (miou, fb_iou, meter).
    Sources, first match wins:
      episodes   -- indexable of HOST episodes (decoded PIL images / uint8 arrays + class-id masks, the
                    material of DatasetCOCO.load_frame, coco.py:77-107): this rank's share goes through the
                    GPU input pipeline (input_pipeline.EpisodeLoader: resize / normalise / mask kernels on a
                    side stream, prefetched);
      make_batch -- make_batch(indices) returns device tensors (dict of episodes.make_episode_batch + 'class_id');
      otherwise  -- synthetic episodes (episodes.make_episode_batch).
    captured: every step is one replay of the pipeline-owned HIP graph (pipeline.run_episodes(captured=True));
    the per-step results are consumed (meter update on the same stream) before the next replay overwrites them.
    r_threshold / threshold / batch_max: the launcher's thresholding flags (main_oss.py:128-135)."""
    device = device or pipe.device
    meter = AverageMeter(benchmark, fold_class_ids(benchmark, fold), device=device)
    mine = ep.shard(n_episodes, rank, world_size)
    if episodes is not None:
        from .input_pipeline import EpisodeLoader
        loader = EpisodeLoader((episodes[i] for i in mine), res, batch, nshot, device=device)
        for bt in loader:
            r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"],
                                  r_threshold=r_threshold, threshold=threshold, batch_max=batch_max, captured=captured)
            meter.update_from_counts(r["counts"], bt["class_id"].to(device))
        meter.all_reduce()
        miou, fb_iou, _ = meter.compute_iou()
        return float(miou), float(fb_iou), meter
    for idx in episode_batches(mine, batch):
        if make_batch is not None:
            bt = make_batch(idx)
            cls = bt["class_id"]
        else:
            bt = ep.make_episode_batch(len(idx), nshot, res, seed=1000 + idx[0], device=device)
            cls = ep.episode_class_ids(idx, benchmark, fold)
        r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"],
                              r_threshold=r_threshold, threshold=threshold, batch_max=batch_max, captured=captured)
        meter.update_from_counts(r["counts"], cls.to(device))
    meter.all_reduce()
    miou, fb_iou, _ = meter.compute_iou()
    return float(miou), float(fb_iou), meter
