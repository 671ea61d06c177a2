"""Average per-dispatch counter values per kernel from a rocprofv3 counter_collection.csv."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pick = sys.argv[2] if len(sys.argv) > 2 else ""
for k, d in agg.items():
    if pick in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
