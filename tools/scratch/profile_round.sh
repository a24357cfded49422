#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the default bench.py command (per-kernel durations)
#   2. separate --pmc passes (HBM bytes: FETCH_SIZE, WRITE_SIZE; SQ instruction / wait counters)
# Outputs under gpurun_out/round/; copy the summaries into profiles/ afterwards
# (tools/profile_collect.py does that).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# FIF=<n>: frames in flight of the bench (default: the bench's own default); output then under gpurun_out/round_fif<n>
OUT=gpurun_out/round${FIF:+_fif$FIF}
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 48 --warmup 16 --no-cpu ${FIF:+--frames-in-flight $FIF}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
[ -n "$FIF" ] && exit 0   # the frames-in-flight variants only need durations and HBM bytes
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1
tail -1 $OUT/stats.log
