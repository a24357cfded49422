#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfg5q
for v in "$@"; do
  env $v python bench.py --tag p11_1080 --samples 64 --paths --steps 2 --warmup 1 --repeats 2 --no-cpu --rays-per-frame 661228579 > gpurun_out/cfg5q/out.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/cfg5q/out.json").read().strip().splitlines()[-1])
ks = d["roofline"]["kernels"]
print("$v", d["ms_per_step"], " ".join("%s=%.0fx%d" % (k, 1000 * v["ms"], v["launches_per_frame"]) for k, v in list(ks.items())[:5]))
PY
done
