"""Summarise a rocprofv3 counter_collection.csv: per kernel and counter -> launches, mean, max (full-size launches = max)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    for c, x in sorted(v.items()):
        print(f"{k:50s} {c:28s} n={len(x):4d} mean={sum(x)/len(x):16.1f} max={max(x):16.1f}")
