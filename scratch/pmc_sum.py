"""Sum a rocprofv3 counter_collection.csv per (kernel, counter): python scratch/pmc_sum.py dir/a_counter_collection.csv [substr]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if sub in r["Kernel_Name"]:
        a = acc[(r["Kernel_Name"][:50], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:50s} {c:28s} n={n:4d} per-dispatch={v / n:16.1f}")
