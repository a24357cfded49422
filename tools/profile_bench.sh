#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the bench command (per-kernel durations)
#   2. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM bytes), two SQ passes (instruction mix, lane utilisation, waits)
# Outputs under gpurun_out/$TAG/; tools/profile_summary.py turns them into the summaries committed under profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r03}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
# (--repeats 2: the profiler serialises and pads every launch; two regions of 64 frames are 4 launch sequences of 32 after the warm-up)
# (--contexts 1: one launch sequence at a time, so that tools/profile_summary.py can tell the levels of the kernels by their order)
B="python3 bench.py --steps 64 --warmup 32 --repeats 2 --no-cpu --contexts 1 $BENCH_ARGS"
echo "$B" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 1
if [ -z "$HBM_ONLY" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || exit 1
fi
tail -1 $OUT/stats.log | cut -c1-300
