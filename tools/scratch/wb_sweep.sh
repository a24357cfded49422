#!/bin/bash
cd $GRAFT_REPO_ROOT
for wb in 0 128 400; do
  export RTU_WALK_BUDGET=$wb
  echo "== RTU_WALK_BUDGET=$wb"
  bash tools/scratch/cfg5_quick.sh
  python bench.py --no-cpu --steps 20 --warmup 5 --repeats 20 > gpurun_out/wbs.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/wbs.json').read().strip().splitlines()[-1]);print('K20', d['value'],d['ms_per_step']);ks=d['roofline']['kernels'];print(' '.join('%s=%.0f'%(k,1000*v['ms']) for k,v in ks.items() if v['ms']>0.1 or 'long' in k))"
  python bench.py --no-cpu --repeats 20 > gpurun_out/wbs.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/wbs.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step']);ks=d['roofline']['kernels'];print(' '.join('%s=%.0f'%(k,1000*v['ms']) for k,v in ks.items() if v['ms']>0.1 or 'long' in k))"
done
