#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 20 > gpurun_out/g20.json 2>gpurun_out/g.err || { tail -3 gpurun_out/g.err; return; }
python bench.py --no-cpu --repeats 20 > gpurun_out/gd.json 2>gpurun_out/g.err || { tail -3 gpurun_out/g.err; return; }
python - "$1" <<PY
import json, sys
a=json.loads(open("gpurun_out/g20.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/gd.json").read().strip().splitlines()[-1])
print("%-28s K20 %.0f  default %.0f  single %.4f" % (sys.argv[1], a["value"], b["value"], a["config"]["single_frame"]["ms_per_frame"]))
PY
}
run base
for v in 8192 16384; do RTU_EXP_GT=$v run "GT=$v"; done
for v in 8192 16384; do RTU_EXP_GN=$v run "GN=$v"; done
for v in 2048 4096 16384; do RTU_EXP_GS=$v run "GS=$v"; done
for v in 4096 8192; do RTU_EXP_GF0=$v run "GF0=$v"; done
for v in 1024 2048; do RTU_EXP_GF=$v run "GF=$v"; done
