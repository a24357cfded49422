#!/bin/bash
cd $GRAFT_REPO_ROOT
for tl in 256 2048 4000; do
RTU_TAIL_LEARN=$tl python bench.py --no-cpu --steps 20 --warmup 5 --repeats 20 > gpurun_out/tl.json 2>gpurun_out/tl.err || { tail -3 gpurun_out/tl.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/tl.json").read().strip().splitlines()[-1]);ks=d["roofline"]["kernels"]
print("tail learn $tl K20:", d["value"], d["ms_per_step"], " ".join("%s=%.0f"%(k,1000*v["ms"]) for k,v in ks.items() if "L3" in k or "L4" in k or "tail" in k))
PY
done
