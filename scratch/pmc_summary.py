"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name (mean per dispatch)."""
import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -sum(cnt[k].values())):
    parts = [f"{c}={agg[k][c] / cnt[k][c]:.4g}" for c in sorted(agg[k])]
    n = max(cnt[k].values())
    print(f"{k:70s} n={n:5d} " + " ".join(parts))
