"""Aggregate rocprofv3 counter CSVs (one --pmc pass per counter, as MI355X_MICROARCH.md prescribes)
into per-kernel totals, then hand them to summarize_pmc.py's format.

On the GPU box (each pass its own process; --pmc is never combined with other trace domains):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o runc -- \
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o runc -- \
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph
    python profiles/collect_pmc.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic_raw.json
    python profiles/summarize_pmc.py gpurun_out/pmc_traffic_raw.json profiles/r01_pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys


def collect(d, out):
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = out.setdefault(row["Kernel_Name"], {})
                c = k.setdefault(row["Counter_Name"], {"dispatches": 0, "sum": 0.0})
                c["dispatches"] += 1
                c["sum"] += float(row["Counter_Value"])


def main(*args):
    *dirs, dst = args
    out = {}
    for d in dirs:
        collect(d, out)
    for k in out.values():
        for c in k.values():
            c["per_dispatch"] = c["sum"] / max(1, c["dispatches"])
    json.dump(out, open(dst, "w"), indent=1)
    print(f"{len(out)} kernels -> {dst}")


if __name__ == "__main__":
    main(*sys.argv[1:])
