#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 32768 65536 131072 262144 524288; do echo "GN=$v"; RTU_EXP_GN=$v bash tools/scratch/cfg5_quick.sh | head -2; done
