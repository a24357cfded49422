#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --tag p11_1080 --samples 16 --paths --steps 1 --warmup 1 --repeats 3 --no-cpu --rays-per-frame 1000000 > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/c5.json").read().strip().splitlines()[-1])
print("cfg5 16 spp ms/frame", d["ms_per_step"])
ks = d["roofline"].get("kernels", {})
print("   ", " ".join("%s=%.0f" % (k, 1000 * v.get("ms", 0)) for k, v in ks.items() if v.get("ms", 0) > 0.004))
PY
