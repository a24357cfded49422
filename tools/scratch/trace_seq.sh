#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_seq; mkdir -p gpurun_out/trace_seq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_seq -- python3 bench.py --no-cpu --contexts 1 --steps ${STEPS:-64} --warmup 32 --repeats 2 $BENCH_ARGS > gpurun_out/trace_seq/log.txt 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/trace_seq/*/*_kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "::k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last sequence of feature set 4 kernels: from the last k_prelude before the last k_combine<4>
idx = [i for i, r in enumerate(rows) if "k_prelude" in r["Kernel_Name"]]
# pick the 3rd last prelude (a timed batched sequence, not the single-frame runs at the end)
cands = [i for i in idx if any("k_primary<" in rows[j]["Kernel_Name"] and ", 4>" in rows[j]["Kernel_Name"] for j in range(i, min(i + 4, len(rows))))]
i0 = cands[-2]
i1 = cands[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("::")[-1].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  q%-3s %-26s start %8.1f end %8.1f dur %7.1f" % (r.get("Queue_Id", "?"), n[:26], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
