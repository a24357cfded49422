#!/bin/bash
cd $GRAFT_REPO_ROOT
for G in ${GRIDS:-1024 1536 2048 3072 4096 6144 8192 16384 32768}; do
export RTU_EXP_PGRID=$G
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 20 > gpurun_out/pg20.json 2>gpurun_out/pg.err || { tail -3 gpurun_out/pg.err; continue; }
python bench.py --no-cpu --repeats 20 > gpurun_out/pgd.json 2>gpurun_out/pg.err || { tail -3 gpurun_out/pg.err; continue; }
python - <<PY
import json
a=json.loads(open("gpurun_out/pg20.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/pgd.json").read().strip().splitlines()[-1])
print("grid $G: K20 %.0f (k_primary %.0f us)  default %.0f (k_primary %.0f us)  single %.4f / %.4f" % (a["value"], 1000*a["roofline"]["kernels"]["k_primary"]["ms"], b["value"], 1000*b["roofline"]["kernels"]["k_primary"]["ms"], a["config"]["single_frame"]["ms_per_frame"], b["config"]["single_frame"]["ms_per_frame"]))
PY
done
