#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  python bench.py --no-cpu --steps 20 --warmup 20 --repeats 30 "$@" > gpurun_out/kv.json 2> gpurun_out/kv.err || { tail -3 gpurun_out/kv.err; return; }
  python - "$*" <<PY
import json, sys
d = json.loads(open("gpurun_out/kv.json").read().strip().splitlines()[-1])
print(sys.argv[1], "->", d["value"], d["ms_per_step"], "region", d["config"]["repeats"]["region_ms_median"], "seqs", d["config"]["launch_sequences_in_flight"], "fif", d["config"]["frames_in_flight"])
PY
}
run
run --frames-in-flight 10 --contexts 2
run --frames-in-flight 10 --contexts 2 --coop-threshold 30000
run --frames-in-flight 10 --contexts 2 --coop-threshold 10000
run --frames-in-flight 10 --contexts 2 --coop-threshold 1
run --frames-in-flight 7 --contexts 3 --coop-threshold 10000
