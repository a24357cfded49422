#!/bin/bash
# kernel trace of the default bench (two contexts alternating): what runs beside what. Prints a window of the timed region.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_two; mkdir -p gpurun_out/trace_two
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_two -- python3 bench.py --no-cpu --steps ${STEPS:-320} --warmup 64 --repeats 2 $BENCH_ARGS > gpurun_out/trace_two/log.txt 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/trace_two/*/*_kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "::k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the middle of the last long run of batched kernels
idx = [i for i, r in enumerate(rows) if "k_prelude" in r["Kernel_Name"]]
mid = idx[len(idx) * 3 // 4]
i0 = mid
i1 = next(i for i in idx if i > mid + 60)
t0 = int(rows[i0]["Start_Timestamp"])
busy = 0
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("::")[-1].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  q%-3s %-26s start %8.1f end %8.1f dur %7.1f" % (r.get("Queue_Id", "?"), n[:26], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
