"""Per-kernel mean of every counter in a rocprofv3 --pmc counter_collection.csv (developer tool)."""
import collections, csv, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in d.items():
    if pat in k:
        print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
