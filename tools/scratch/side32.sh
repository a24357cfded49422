#!/bin/bash
cd $GRAFT_REPO_ROOT
RTU_SIDE_VERBOSE=1 python bench.py --no-cpu --contexts 1 --steps 64 --warmup 32 --repeats 3 > gpurun_out/s32.json 2> gpurun_out/s32.err
grep -c "\[side\]" gpurun_out/s32.err; grep "\[side\]" gpurun_out/s32.err | tail -2
python -c "
import json;d=json.loads(open('gpurun_out/s32.json').read().strip().splitlines()[-1]); print('ctx1 32 frames', d['value'], d['ms_per_step']); print(' '.join(k for k in d['roofline']['kernels']))"
for q in 4 8; do
for c in 1 2 3; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --contexts $c --repeats 20 > gpurun_out/s32.json 2> gpurun_out/s32.err
python -c "
import json;d=json.loads(open('gpurun_out/s32.json').read().strip().splitlines()[-1]); print('queues $q contexts $c:', d['value'], d['ms_per_step'], 'side' if 'k_tail(side)' in d['roofline']['kernels'] else '')"
done
done
