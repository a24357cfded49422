#!/bin/bash
cd $GRAFT_REPO_ROOT
for sm in 256 4000; do
for c in 1 2 3; do
RTU_SIDE_MAX=$sm python bench.py --no-cpu --contexts $c --repeats 20 > gpurun_out/s32.json 2> gpurun_out/s32.err || { tail -3 gpurun_out/s32.err; continue; }
python -c "
import json;d=json.loads(open('gpurun_out/s32.json').read().strip().splitlines()[-1]); print('side max $sm contexts $c:', d['value'], d['ms_per_step'], 'side' if 'k_tail(side)' in d['roofline']['kernels'] else '', d['config']['z_bit_exact_vs_reference_golden'])"
done
done
