#!/bin/bash
# Is the one-lane-per-ray BVH walk bound by the texture-addresser / L1 path? TA busy cycles per kernel (two counters per pass).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ta
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 32 --warmup 16 --no-cpu"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1
tail -n 2 $OUT/p1.log $OUT/p2.log $OUT/p3.log
