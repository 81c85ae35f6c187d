"""Generate tests/golden/classid_goldens.json by IMPORTING the reference's dataset classes
(evaluation_util/data/{coco,pascal,fss}.py import only torch / PIL / numpy) and calling their
`build_class_ids` on a stub carrying the attributes the method reads -- no dataset files are touched.

Run once in the build container:  python tests/golden/make_classid_goldens.py
The reference never travels; only the resulting inputs/outputs are committed.
"""
import importlib.util
import json
import os
import sys
from types import SimpleNamespace

REF = "/root/reference/evaluation_util/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "classid_goldens.json")


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_data_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    out = []
    coco, pascal, fss = _load("coco").DatasetCOCO, _load("pascal").DatasetPASCAL, _load("fss").DatasetFSS
    for fold in range(4):
        for split in ("val", "trn"):
            out.append(dict(benchmark="coco", fold=fold, split=split,
                            ids=list(coco.build_class_ids(SimpleNamespace(nclass=80, nfolds=4, fold=fold, split=split)))))
            out.append(dict(benchmark="pascal", fold=fold, split=split,
                            ids=list(pascal.build_class_ids(SimpleNamespace(nclass=20, nfolds=4, fold=fold, split=split)))))
    for split in ("trn", "val", "test"):
        out.append(dict(benchmark="fss", fold=0, split=split, ids=list(fss.build_class_ids(SimpleNamespace(split=split)))))
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, len(out), "cases")


if __name__ == "__main__":
    sys.exit(main())
