#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per dispatch, per kernel name.
Only the fast (non-counting) kernel variants of the timed frames are kept."""
import csv, glob, sys, collections, re
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc*/*/*_counter_collection.csv"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(pat)):
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "::k_" not in k or "true>" in k:
            continue
        name = k.split("::")[-1].split("(")[0]
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = name
    # number the dispatches of each kernel within a frame (level order)
    seq = collections.defaultdict(int)
    order = {}
    for d in sorted(names, key=int):
        n = names[d]
        if n.startswith("k_primary"):
            seq.clear()
        order[d] = "%s#%d" % (n, seq[n])
        seq[n] += 1
    for (d, c), v in per.items():
        agg[order[d]][c].append(v)
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-26s n=%2d mean=%.6g" % (c, len(v), sum(v) / len(v)))
