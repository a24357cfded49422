#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03f
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03f/prof -- python3 bench.py --tag p11_1080 --samples 64 --paths --steps 1 --warmup 1 --repeats 1 --no-cpu $CFG5_ARGS > gpurun_out/r03f/prof.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/r03f/prof/*/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(r["Name"].split("::")[-1][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
