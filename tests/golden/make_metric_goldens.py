"""Generate tests/golden/metric_goldens.json by IMPORTING the reference's own
evaluation_util/common/{evaluation,logger}.py (the only hot-path-adjacent
reference files importable in the build container).

Run once in the build container:  python tests/golden/make_metric_goldens.py
The reference never travels; only the resulting inputs/outputs are committed.
"""
import importlib.util
import json
import os
import sys

import torch

REF = "/root/reference/evaluation_util/common"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metric_goldens.json")


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self  # GPU-less box: AverageMeter calls .cuda()
    ev = _load("evaluation")
    lg = _load("logger")
    ev.Evaluator.initialize()
    cases = []
    g = torch.Generator().manual_seed(7)
    for case_id, (b, hw, with_ignore) in enumerate([(2, 64, False), (3, 32, True), (1, 16, False), (4, 8, True)]):
        pred = (torch.rand(b, hw, hw, generator=g) > 0.5).float()
        gt = (torch.rand(b, hw, hw, generator=g) > 0.6).float()
        batch = {"query_mask": gt.clone()}
        ignore = None
        if with_ignore:
            ignore = ((torch.rand(b, hw, hw, generator=g) > 0.9) & (gt == 0)).float()
            batch["query_ignore_idx"] = ignore.clone()
        inter, union = ev.Evaluator.classify_prediction(pred.clone(), batch)
        cases.append(dict(case=case_id, pred=pred.int().tolist(), gt=gt.int().tolist(),
                          ignore=None if ignore is None else ignore.int().tolist(),
                          inter=inter.tolist(), union=union.tolist()))

    # AverageMeter over a synthetic COCO fold-0 stream
    class _DS:
        benchmark = "coco"
        class_ids = [0 + 4 * v for v in range(20)]  # coco.py:64-70, fold 0 val
    meter = lg.AverageMeter(_DS())
    stream = []
    for i in range(40):
        inter = torch.randint(0, 5000, (2, 1), generator=g).float()
        union = inter + torch.randint(1, 5000, (2, 1), generator=g).float()
        cid = torch.tensor([_DS.class_ids[i % 20]])
        meter.update(inter, union, cid, loss=None)
        stream.append(dict(inter=inter.tolist(), union=union.tolist(), class_id=cid.tolist()))
    miou, fb_iou, _ = meter.compute_iou()
    out = dict(classify=cases,
               meter=dict(benchmark="coco", class_ids=_DS.class_ids, stream=stream,
                          miou=float(miou), fb_iou=float(fb_iou),
                          intersection_buf=meter.intersection_buf.tolist(), union_buf=meter.union_buf.tolist()))
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
