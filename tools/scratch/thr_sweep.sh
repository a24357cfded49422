#!/bin/bash
cd $GRAFT_REPO_ROOT
for thr in 70000 110000 160000; do
python bench.py --no-cpu --steps 20 --warmup 5 --repeats 20 --coop-threshold $thr > gpurun_out/ts.json 2>gpurun_out/ts.err || { tail -3 gpurun_out/ts.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/ts.json").read().strip().splitlines()[-1]);ks=d["roofline"]["kernels"]
print("thr $thr K20:", d["value"], d["ms_per_step"], "single", d["config"]["single_frame"]["ms_per_frame"], " ".join("%s=%.0f"%(k,1000*v["ms"]) for k,v in ks.items() if "trace2" in k))
PY
python bench.py --no-cpu --repeats 20 --coop-threshold $thr > gpurun_out/ts.json 2>gpurun_out/ts.err || { tail -3 gpurun_out/ts.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/ts.json").read().strip().splitlines()[-1]);ks=d["roofline"]["kernels"]
print("thr $thr default:", d["value"], d["ms_per_step"], " ".join("%s=%.0f"%(k,1000*v["ms"]) for k,v in ks.items() if "trace2" in k))
PY
done
