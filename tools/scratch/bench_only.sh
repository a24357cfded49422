#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-b}; mkdir -p gpurun_out/$TAG
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/$TAG/bench20.json 2> gpurun_out/$TAG/bench20.err || { tail -5 gpurun_out/$TAG/bench20.err; exit 1; }
python bench.py --no-cpu > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err || { tail -5 gpurun_out/$TAG/bench_default.err; exit 1; }
python - <<PY
import json
for f in ("bench20", "bench_default"):
    d = json.loads(open("gpurun_out/$TAG/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "Mrays/s", d["value"], "ms/frame", d["ms_per_step"], "single", d["config"].get("single_frame", {}).get("ms_per_frame"), "z ok", d["config"]["z_bit_exact_vs_reference_golden"])
    ks = d["roofline"].get("kernels", {})
    print("   ", " ".join("%s=%.0f" % (k, 1000 * v.get("ms", 0)) for k, v in ks.items() if v.get("ms", 0) > 0.004))
PY
