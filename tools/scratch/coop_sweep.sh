#!/bin/bash
cd $GRAFT_REPO_ROOT
for thr in 1 4000 20000; do
  for c in 2; do
    python bench.py --no-cpu --contexts $c --coop-threshold $thr --repeats 20 > gpurun_out/cs.json 2> gpurun_out/cs.err || { tail -3 gpurun_out/cs.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/cs.json").read().strip().splitlines()[-1])
print("thr $thr contexts $c:", d["value"], d["ms_per_step"], "single", d["config"]["single_frame"]["ms_per_frame"])
PY
  done
done
python bench.py --no-cpu --steps 20 --warmup 5 --coop-threshold 4000 > gpurun_out/cs.json 2> gpurun_out/cs.err
python - <<PY
import json
d = json.loads(open("gpurun_out/cs.json").read().strip().splitlines()[-1])
print("thr 4000 steps 20:", d["value"], d["ms_per_step"], "single", d["config"]["single_frame"]["ms_per_frame"])
PY
