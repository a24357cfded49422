#!/usr/bin/env python3
"""Print the kernel timeline of the LAST frame found in a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof*/*/*_kernel_trace.csv"
for f in sorted(glob.glob(pat)):
    rows = [r for r in csv.DictReader(open(f)) if "::k_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_primary<" in r["Kernel_Name"]]
    if not idx:
        continue
    last = rows[idx[-1]:]
    t0 = int(last[0]["Start_Timestamp"])
    print(f)
    for r in last:
        n = r["Kernel_Name"].split("::")[-1].split("(")[0]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  %-28s start %8.1f us  dur %8.1f us  vgpr %s" % (n, (s - t0) / 1e3, (e - s) / 1e3, r.get("VGPR_Count", "?")))
    print("  frame span %.1f us" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3))
