#!/bin/bash
cd $GRAFT_REPO_ROOT
for dbg in "$@"; do
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 10 --dbg $dbg > gpurun_out/ds.json 2>gpurun_out/ds.err || { tail -3 gpurun_out/ds.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/ds.json").read().strip().splitlines()[-1]);ks=d["roofline"]["kernels"]
print("dbg $dbg:", d["ms_per_step"], " ".join("%s=%.0f"%(k,1000*v["ms"]) for k,v in ks.items() if v["ms"]>0.03))
PY
done
