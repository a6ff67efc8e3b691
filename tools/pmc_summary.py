#!/usr/bin/env python3
"""tools/pmc_summary.py <dir> -- per-kernel means of the counters in a rocprofv3 --pmc output tree (counter_collection.csv)"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if not k.startswith("k_") and "k_" not in k: continue
        k = (k[k.index("k_"):] if "k_" in k else k).replace(", ", ";").replace(",", ";")
        per[(row["Dispatch_Id"], k, row["Counter_Name"])] += float(row["Counter_Value"])
    for (d, k, c), v in per.items(): acc[k][c].append(v)
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names) + ",dispatches")
for k in sorted(acc):
    if k in ("k_repack_occ", "k_repack_hash", "k_build_gen2", "k_expand_sa", "k_build_t20") or k.startswith("k_ib_"): continue
    n = max(len(v) for v in acc[k].values())
    print(k + "," + ",".join("%.0f" % (sum(acc[k][c][-2:]) / max(1, len(acc[k][c][-2:]))) if c in acc[k] else "" for c in names) + ",%d" % n)
