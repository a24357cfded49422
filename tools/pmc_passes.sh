set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc1 -- $B > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc2 -- $B > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc3 -- $B > gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc4 -- $B > gpurun_out/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD --output-format csv -d gpurun_out/pmc5 -- $B > gpurun_out/pmc5.log 2>&1
ls gpurun_out/pmc*/*/ | head -40
