#!/bin/bash
# One GPU call of the edit-measure loop: the parity suite, then the bench with the driver's arguments and its default.
# usage (through gpurun): bash tools/quick_gpu.sh <tag> [pytest args]
cd $GRAFT_REPO_ROOT
TAG=${1:-quick}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1
RC=$?
tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench20.json 2> $OUT/bench20.err || { tail -5 $OUT/bench20.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - <<PY
import json
for f in ("bench20", "bench_default"):
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "Mrays/s", d["value"], "ms/frame", d["ms_per_step"], "single", d["config"].get("single_frame", {}).get("ms_per_frame"))
    ks = d["roofline"].get("kernels", {})
    if isinstance(ks, dict):
        print("   ", " ".join("%s=%.0f" % (k, 1000 * v.get("ms", 0)) for k, v in ks.items() if v.get("ms", 0) > 0.004))
PY
