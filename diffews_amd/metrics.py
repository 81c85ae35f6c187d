"""Episode metric: intersection/union accumulation and mIoU / FB-IoU, plus the multi-GPU reduction.

Counterpart of `Evaluator.classify_prediction`
(/root/reference/evaluation_util/common/evaluation.py:12-39) and `AverageMeter`
(/root/reference/evaluation_util/common/logger.py:12-51).  The per-episode counts come from the
on-device kernel `dfw_seg_postprocess` (or from `classify_prediction` below for ready-made masks).

Deliberate difference: the reference accumulates pixel counts in fp32 `[2, nclass]` buffers, exact
only below 2**24 per cell; one COCO class cell receives ~50 episodes x 262144 px.  Here the buffers
are int64 (order-independent => bit-identical for any GPU count) and are converted to fp32 only inside
`compute_iou`, which then follows logger.py:42-51 literally.  The all-reduce of the two buffers is the
only collective on the evaluation path: one sum-all-reduce per evaluation.
"""
import torch

NCLASS = {"pascal": 20, "coco": 80, "fss": 1000, "paco_part": 448, "pascal_part": 100, "lvis": 1203}


def fold_class_ids(benchmark, fold, nfolds=None, split="val", cat_ids=None):
    """Classes a dataset split evaluates / trains on (`dataset.class_ids`, logger.py:15), per benchmark:
      coco   (evaluation_util/data/coco.py:64-70)    interleaved: val = fold + nfolds*v, v < nclass/nfolds
      pascal (evaluation_util/data/pascal.py:115-123) contiguous:  val = fold*n + i,     i < nclass/nfolds
      fss    (evaluation_util/data/fss.py:100-107)    fixed ranges per split, no folds: trn 0..519,
             val 520..759, test 760..999
      lvis   (evaluation_util/data/lvis.py:66-90)     10 folds, interleaved over the annotation file's
             category list -- pass that list as `cat_ids` (it comes from lvis_val.pkl / lvis_train.pkl,
             not from arithmetic); the meter is then indexed by position (lvis.py:28-29).
    Other benchmarks (paco_part, pascal_part) keep their class lists in dataset pickles: not derivable."""
    nclass = NCLASS[benchmark]
    if benchmark == "coco":
        nf = nfolds or 4
        val = [fold + nf * v for v in range(nclass // nf)]
        return val if split != "trn" else [c for c in range(nclass) if c not in val]
    if benchmark == "pascal":
        nf = nfolds or 4
        n = nclass // nf
        val = [fold * n + i for i in range(n)]
        return val if split != "trn" else [c for c in range(nclass) if c not in val]
    if benchmark == "fss":
        return {"trn": list(range(0, 520)), "val": list(range(520, 760)), "test": list(range(760, 1000))}[split]
    if benchmark == "lvis":
        if cat_ids is None:
            raise ValueError("lvis class ids come from the annotation pickles (lvis.py:66-90): pass cat_ids")
        nf = nfolds or 10
        cats = list(cat_ids)
        if split == "trn":
            raise NotImplementedError("lvis training split: class list = train categories minus the fold's "
                                      "validation categories (lvis.py:84), both from the annotation pickles")
        picked = [cats[fold + nf * v] for v in range(len(cats) // nf)]
        return sorted(range(len(picked)))
    raise NotImplementedError(f"class ids of benchmark {benchmark!r} live in its dataset files "
                              "(evaluation_util/data/%s.py); pass them to AverageMeter directly" % benchmark)


def classify_prediction(pred_mask, gt_mask, query_ignore_idx=None, ignore_index=255):
    """pred/gt [B, H, W] with values {0, 1}; -> (area_inter [2, B], area_union [2, B]) int64.
    Integer restatement of evaluation.py:12-39 (torch.histc with 2 bins over [0, 1] drops 255)."""
    pred = pred_mask.long()
    gt = gt_mask.long()
    if query_ignore_idx is not None:
        ign = query_ignore_idx.bool()
        gt = torch.where(ign, torch.full_like(gt, ignore_index), gt)
        pred = torch.where(ign, torch.full_like(pred, ignore_index), pred)
    out_i, out_u = [], []
    for p, g in zip(pred, gt):
        inter = torch.stack([((p == c) & (g == c)).sum() for c in (0, 1)])
        pa = torch.stack([(p == c).sum() for c in (0, 1)])
        ga = torch.stack([(g == c).sum() for c in (0, 1)])
        out_i.append(inter)
        out_u.append(pa + ga - inter)
    return torch.stack(out_i).t().contiguous(), torch.stack(out_u).t().contiguous()


class AverageMeter:
    def __init__(self, benchmark, class_ids, device="cpu"):
        self.benchmark = benchmark
        self.nclass = NCLASS[benchmark]
        self.class_ids_interest = torch.tensor(list(class_ids), dtype=torch.long, device=device)
        self.intersection_buf = torch.zeros(2, self.nclass, dtype=torch.int64, device=device)
        self.union_buf = torch.zeros(2, self.nclass, dtype=torch.int64, device=device)

    def update(self, inter_b, union_b, class_id):
        """inter_b / union_b [2, B] counts, class_id [B] (logger.py:35-37)."""
        cid = class_id.to(self.intersection_buf.device, torch.long)
        self.intersection_buf.index_add_(1, cid, inter_b.to(self.intersection_buf.device, torch.int64))
        self.union_buf.index_add_(1, cid, union_b.to(self.union_buf.device, torch.int64))

    def update_from_counts(self, counts, class_id):
        """counts [B, 4] int64 = inter0, inter1, union0, union1 from dfw_seg_postprocess.  On the GPU this
        is one library kernel (dfw_meter_update: int64 atomics, exact in any order); CPU buffers (the gloo
        rehearsal, host-side checks) take the index_add_ form of logger.py:35-37."""
        if self.intersection_buf.is_cuda:
            from . import ops
            cid = class_id.to(self.intersection_buf.device, torch.int64).contiguous()
            ops.meter_update(counts.contiguous(), cid, self.intersection_buf, self.union_buf)
            return
        self.update(counts[:, 0:2].t(), counts[:, 2:4].t(), class_id)

    def all_reduce(self, group=None):
        """Sum the two buffers over all ranks (RCCL over xGMI when the buffers are on the GPU,
        gloo on CPU).  int64 sums are exact and order-independent."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            both = torch.stack([self.intersection_buf, self.union_buf])
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
            self.intersection_buf, self.union_buf = both[0].clone(), both[1].clone()

    def compute_iou(self):
        """logger.py:42-51 on fp32 copies of the (exact) buffers."""
        inter, union = self.intersection_buf.float(), self.union_buf.float()
        iou = inter / torch.max(torch.stack([union, torch.ones_like(union)]), dim=0)[0]
        iou = iou.index_select(1, self.class_ids_interest)
        miou = iou[1].mean() * 100
        fb_iou = (inter.index_select(1, self.class_ids_interest).sum(dim=1)
                  / union.index_select(1, self.class_ids_interest).sum(dim=1)).mean() * 100
        return miou, fb_iou, iou[1][:min(len(iou[1]), 20)]
