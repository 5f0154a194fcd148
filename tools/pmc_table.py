"""Pivot a rocprofv3 counter_collection.csv: per kernel, mean of each counter over its dispatches."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({r["Counter_Name"] for r in rows})
print("| kernel | n | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", kv[1].get(names[0], [0])))):
    mm = re.search(r"(\w+_kernel)(<[^>]*>|I[^E]*E)?", k)
    short = (mm.group(0) if mm else k)[:60]
    n = len(next(iter(d.values())))
    print("| `%s` | %d | " % (short, n) + " | ".join("%.4g" % (sum(d[c]) / len(d[c])) if c in d else "-" for c in names) + " |")
