#!/bin/bash
# rocprofv3 PMC passes over bench.py (one pass per counter group; --pmc never combined
# with other trace domains). Usage: tools/pmc_passes.sh <golden tag> <out prefix>
TAG=${1:-teapot2_1080}
OUT=${2:-gpurun_out/pmc}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 2 --no-cpu --tag $TAG"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d ${OUT}1 -- $B > ${OUT}1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d ${OUT}2 -- $B > ${OUT}2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d ${OUT}3 -- $B > ${OUT}3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d ${OUT}4 -- $B > ${OUT}4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT --output-format csv -d ${OUT}5 -- $B > ${OUT}5.log 2>&1
