#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/scratch/cfg5_quick.sh | head -2
for v in 8192 16384 65536; do echo "GN=$v"; RTU_EXP_GN=$v bash tools/scratch/cfg5_quick.sh | head -2; done
for v in 4096 8192 16384; do echo "PS=$v"; RTU_EXP_PS=$v bash tools/scratch/cfg5_quick.sh | head -2; done
