#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 50 > gpurun_out/g20.json 2>gpurun_out/g.err || { tail -3 gpurun_out/g.err; return; }
python - "$1" <<PY
import json, sys
a=json.loads(open("gpurun_out/g20.json").read().strip().splitlines()[-1])
print("%-24s K20 %.0f  (region min %.4f)" % (sys.argv[1], a["value"], a["config"]["repeats"]["region_ms_min"]))
PY
}
for rep in 1 2 3; do
run base
RTU_EXP_GT=16384 run "GT=16384"
RTU_EXP_GN=16384 run "GN=16384"
RTU_EXP_GT=16384 RTU_EXP_GN=16384 run "GT=GN=16384"
RTU_EXP_GT=16384 RTU_EXP_GN=16384 RTU_EXP_GS=4096 run "GT=GN=16384 GS=4096"
done
