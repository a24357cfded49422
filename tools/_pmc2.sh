cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmcm_
B="python bench.py --steps 4 --warmup 2 --no-cpu --tag ${1:-teapot2_1080}"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d ${OUT}$i -- $B > ${OUT}$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py "gpurun_out/pmcm_*/*/*_counter_collection.csv" > gpurun_out/pmcm_summary.txt
