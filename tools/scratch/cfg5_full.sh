#!/bin/bash
cd $GRAFT_REPO_ROOT
for lg in ${LOGS:-25 26 27}; do
RTU_GI_BATCH_LOG2=$lg python bench.py --tag p11_1080 --samples 64 --paths --steps 1 --warmup 1 --repeats 3 --no-cpu --rays-per-frame 1000000 > gpurun_out/c5f.json 2> gpurun_out/c5f.err || { tail -3 gpurun_out/c5f.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/c5f.json").read().strip().splitlines()[-1])
print("log2 $lg: cfg5 64 spp ms/frame", d["ms_per_step"], d["config"].get("repeats", {}).get("region_ms_min"))
PY
done
