"""Print the mean per-dispatch PMC counters of the SA kernel for one or more scripts/profile_ab.sh output directories side by side."""
import collections, csv, glob, os, sys
cols = []
for src in sys.argv[1:]:
    vals = {}
    for p in sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))):
        for r in csv.DictReader(open(p)):
            if "sat_sa_kernel" in r["Name"]:
                vals["kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
    for p in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if "sat_sa_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
    cols.append(vals)
keys = []
for c in cols:
    for k in c:
        if k not in keys:
            keys.append(k)
print(f"{'counter':28s}" + "".join(f"{os.path.basename(s.rstrip('/')):>16s}" for s in sys.argv[1:]))
for k in keys:
    print(f"{k:28s}" + "".join(f"{c.get(k, float('nan')):16.6g}" for c in cols))
