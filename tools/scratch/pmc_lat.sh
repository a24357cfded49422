#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmclat; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 64 --warmup 32 --repeats 2 --no-cpu --contexts 1"
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
def load(d):
    f = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        if ", 4>" not in k and "<4>" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k]+=1
    return acc, n
a1,n1 = load("p1"); a2,n2 = load("p2")
for k in sorted(a1, key=lambda k:-a1[k]["SQ_WAVE_CYCLES"])[:12]:
    c=a1[k]; d=a2.get(k,{})
    vm = c["SQ_INSTS_VMEM_RD"]+c["SQ_INSTS_VMEM_WR"]
    print("%-28s vmem lat %7.0f cyc (n/wave %6.1f)  smem lat %6.0f cyc (n/wave %6.1f)  vmem-level/wavecycles %.2f smem-level/wavecycles %.2f | ifetch level/wavecycles %.3f ifetch/wave %.0f  lds lat %.0f wait_lds/wavecycles %.3f" % (
        k[:28], c["SQ_INST_LEVEL_VMEM"]/max(vm,1), vm/max(c["SQ_WAVES"],1), c["SQ_INST_LEVEL_SMEM"]/max(c["SQ_INSTS_SMEM"],1), c["SQ_INSTS_SMEM"]/max(c["SQ_WAVES"],1),
        c["SQ_INST_LEVEL_VMEM"]/max(c["SQ_WAVE_CYCLES"],1), c["SQ_INST_LEVEL_SMEM"]/max(c["SQ_WAVE_CYCLES"],1),
        d.get("SQ_IFETCH_LEVEL",0)/max(d.get("SQ_WAVE_CYCLES",1),1), d.get("SQ_IFETCH",0)/max(c["SQ_WAVES"],1)*n1[k]/max(n2.get(k,1),1),
        d.get("SQ_INST_LEVEL_LDS",0)/max(d.get("SQ_INSTS_LDS",1),1), d.get("SQ_WAIT_INST_LDS",0)/max(d.get("SQ_WAVE_CYCLES",1),1)))
PY
