#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 2 3 4; do
for st in 320 640; do
python bench.py --no-cpu --contexts $c --steps $st --repeats 20 > gpurun_out/cx.json 2>gpurun_out/cx.err || { tail -3 gpurun_out/cx.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/cx.json").read().strip().splitlines()[-1])
print("contexts $c steps $st:", d["value"], d["ms_per_step"], d["config"]["launch_sequences_in_flight"])
PY
done
done
