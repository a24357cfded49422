#!/bin/bash
cd $GRAFT_REPO_ROOT
for tag in teapot2_1080 p4_1080 p11_1080 p3s_800x600; do
for G in 1024 2048 4096 8192 32768; do
export RTU_EXP_PGRID=$G
python bench.py --no-cpu --tag $tag --contexts 1 --steps 64 --warmup 32 --repeats 10 > gpurun_out/ps.json 2>gpurun_out/ps.err || { tail -3 gpurun_out/ps.err; continue; }
python - <<PY
import json
a=json.loads(open("gpurun_out/ps.json").read().strip().splitlines()[-1])
print("$tag grid $G: %.0f Mrays/s  %.4f ms/frame (k_primary %.0f us)  single %.4f" % (a["value"], a["ms_per_step"], 1000*a["roofline"]["kernels"]["k_primary"]["ms"], a["config"]["single_frame"]["ms_per_frame"]))
PY
done
done
