#!/bin/bash
cd $GRAFT_REPO_ROOT
for sp in 6 3 1.5 12; do
export RTU_LGRID_SPAN=$sp
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 20 > gpurun_out/sp.json 2>gpurun_out/sp.err || { tail -3 gpurun_out/sp.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/sp.json").read().strip().splitlines()[-1]);ks=d["roofline"]["kernels"]
print("span $sp:", d["value"], d["ms_per_step"], "single", d["config"]["single_frame"]["ms_per_frame"], " ".join("%s=%.0f"%(k,1000*v["ms"]) for k,v in ks.items() if v["ms"]>0.06))
PY
done
