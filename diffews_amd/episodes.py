"""Synthetic few-shot episodes: the datasets of the reference
(/root/reference/evaluation_util/data/*) are not available offline, so benchmarks and tests draw
episodes with the same tensor contract as `DatasetCOCO.__getitem__`
(evaluation_util/data/coco.py:32-60) after the launcher's mask expansion
(evaluation_util/main_oss.py:100-104): images in [-1, 1], masks exactly +-1 on 3 equal channels,
support tensors batch-major `episode*nshot + shot`.
"""
import torch

from .metrics import fold_class_ids


def make_episode_batch(b, nshot, res, seed=0, device="cpu"):
    """-> dict(support_imgs [b*s,3,H,W], query_img [b,3,H,W], support_masks [b*s,3,H,W] (+-1),
    query_mask uint8 [b,H,W] (0/1))."""
    g = torch.Generator().manual_seed(seed)
    H = W = res
    sup = torch.rand(b * nshot, 3, H, W, generator=g) * 2 - 1
    qry = torch.rand(b, 3, H, W, generator=g) * 2 - 1

    def rect_mask(n):
        m = torch.zeros(n, H, W)
        m[:, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1          # centred rectangle, 25 % area
        speck = (torch.rand(n, H, W, generator=g) < 0.02).float()
        return (m + speck) % 2                                   # XOR Bernoulli(0.02) speckle

    sm = rect_mask(b * nshot)
    masks = sm[:, None].repeat(1, 3, 1, 1) * 2 - 1               # E:100
    qm = rect_mask(b).to(torch.uint8)
    out = dict(support_imgs=sup, query_img=qry, support_masks=masks, query_mask=qm)
    return {k: v.to(device) for k, v in out.items()}


def episode_class_ids(indices, benchmark="coco", fold=0):
    """class id of episode i: i mod 20 mapped into the fold's validation classes."""
    ids = fold_class_ids(benchmark, fold)
    return torch.tensor([ids[i % len(ids)] for i in indices], dtype=torch.long)


def shard(n_episodes, rank, world_size):
    """Static round-robin episode sharding: rank r takes episodes i == r (mod R)."""
    return list(range(rank, n_episodes, world_size))
