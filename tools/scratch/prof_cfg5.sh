#!/bin/bash
# PMC view of config 5's timed kernels (recipe P, 16 samples = one batch of 2^25 chains x 5 depths)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/cfg5p
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --tag p11_1080 --samples 16 --paths --steps 1 --warmup 1 --repeats 1 --no-cpu --rays-per-frame 1000000"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1 || { tail -5 $OUT/pmc_sq1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || { tail -5 $OUT/pmc_sq2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
def load(d):
    f = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    return acc, n
a1, n1 = load("pmc_sq1"); a2, n2 = load("pmc_sq2")
for k in sorted(a1, key=lambda k: -a1[k]["SQ_BUSY_CYCLES"]):
    c = a1[k]; d = a2.get(k, {})
    if c["SQ_WAVES"] < 1 or "k_" not in k: continue
    valu = c["SQ_INSTS_VALU"]; thr = d.get("SQ_THREAD_CYCLES_VALU", 0); act = d.get("SQ_ACTIVE_INST_VALU", 0)
    lanes = thr / (act * 64) if act else 0  # as tools/profile_summary.py
    print("%-34s launches %4d waves/launch %9.0f VALU/wave %7.0f SALU/wave %7.0f VMEM/wave %6.1f LDS/wave %6.1f lanes %.2f wait %.2f" % (
        k[:34], n1[k], c["SQ_WAVES"] / n1[k], valu / c["SQ_WAVES"], c["SQ_INSTS_SALU"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], c["SQ_INSTS_LDS"] / c["SQ_WAVES"], lanes,
        d.get("SQ_WAIT_ANY", 0) / max(1.0, a1[k]["SQ_WAVE_CYCLES"])))
PY
