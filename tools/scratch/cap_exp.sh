#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 5 --dbg 8192 > gpurun_out/cap.json 2>gpurun_out/cap.err || { tail -3 gpurun_out/cap.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/cap.json").read().strip().splitlines()[-1]);ks=d["roofline"]["kernels"]
print("ms/frame", d["ms_per_step"], " ".join("%s=%.0f"%(k,1000*v["ms"]) for k,v in ks.items() if v["ms"]>0.05))
PY
python3 tools/scratch/touched_dump.py teapot2_1080 4 2>&1 | grep -E "k_primary2 |k_trace2\(L[012]\)" | cut -c1-400
