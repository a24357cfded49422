#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 300 python bench.py --no-cpu --repeats 6 "$@" > gpurun_out/rg.json 2> gpurun_out/rg.err || { tail -3 gpurun_out/rg.err; return; }
  python - "$*" <<PY
import json, sys
d = json.loads(open("gpurun_out/rg.json").read().strip().splitlines()[-1]); r = d["config"]["repeats"]
print(sys.argv[1], "-> %.0f Mrays/s, region ms: first %.2f min %.2f median %.2f max %.2f" % (d["value"], r["region_ms_first"], r["region_ms_min"], r["region_ms_median"], r["region_ms_max"]))
PY
}
run --contexts 1 --frames-in-flight 24 --steps 312
run --contexts 1 --frames-in-flight 24 --steps 320
run --contexts 1 --frames-in-flight 32 --steps 328
run --contexts 1 --frames-in-flight 32 --steps 340
run --contexts 2 --frames-in-flight 32 --steps 328
run --contexts 1 --frames-in-flight 24 --steps 320 --dbg 8192
