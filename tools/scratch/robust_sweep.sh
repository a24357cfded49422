#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 300 python bench.py --no-cpu --repeats 5 "$@" > gpurun_out/rs.json 2> gpurun_out/rs.err || { echo "$* FAILED: $(tail -2 gpurun_out/rs.err | tr '\n' ' ')"; return; }
  python - "$*" <<PY
import json, sys
d = json.loads(open("gpurun_out/rs.json").read().strip().splitlines()[-1]); r = d["config"]["repeats"]
flag = "  <-- SPREAD" if r["region_ms_max"] > 1.5 * r["region_ms_min"] else ""
print("%-62s %8.0f Mrays/s  region ms min %.2f median %.2f max %.2f  z %s%s" % (sys.argv[1], d["value"], r["region_ms_min"], r["region_ms_median"], r["region_ms_max"], d["config"].get("z_bit_exact_vs_reference_golden"), flag))
PY
}
for f in 2 3 5 8 12 48 64 100 128; do run --frames-in-flight $f --steps 200 --contexts 1; done
for f in 5 12 48; do run --frames-in-flight $f --steps 203 --contexts 2; done
run --tag p4_1080 --frames-in-flight 24 --steps 100 --contexts 2
run --tag p11_1080 --frames-in-flight 20 --steps 90 --contexts 1
run --tag p3s_800x600 --frames-in-flight 100 --steps 500 --contexts 2
run --tag p7_200x150 --frames-in-flight 16 --steps 100 --contexts 1
run --size 1001x701 --frames-in-flight 20 --steps 50 --contexts 1
run --size 3840x2160 --frames-in-flight 8 --steps 24 --contexts 1
