#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per dispatch of render_kernel<*, false>."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
agg = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/pmc*/*/*_counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r["Kernel_Name"] or "true" in r["Kernel_Name"]:
            continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        agg[c].append(v)
for c, v in sorted(agg.items()):
    print("%-28s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
