#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
for t in 70000 150000 300000 600000 1200000; do
  for steps in 20 320; do
    python bench.py --steps $steps --warmup 5 --no-cpu --coop-threshold $t --repeats 20 > gpurun_out/sweep/c${t}_$steps.json 2>/dev/null
    python - <<PY
import json
d = json.loads(open("gpurun_out/sweep/c${t}_$steps.json").read().strip().splitlines()[-1])
ks = d["roofline"]["kernels"]
print("thr $t steps $steps:", d["value"], d["ms_per_step"], " ".join("%s=%.0f" % (k, 1000 * v["ms"]) for k, v in ks.items() if "trace2" in k or "primary2" in k))
PY
  done
done
