#!/bin/bash
# On the GPU box (through gpurun): GPU parity suite, GPU-clock timeline (TL_TAGS), bench of the four full-size scenes.
# quick GPU check: parity suite + timeline + bench of the four full-size scenes
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 300 python tools/gpu_timeline.py ${TL_TAGS:-teapot2_1080} > gpurun_out/tl.txt 2>&1
for t in teapot2_1080 p11_1080 p4_1080 p3s_800x600; do timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --tag $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$t\", d[\"ms_per_step\"], d[\"value\"], d[\"config\"][\"z_bit_exact_vs_reference_golden\"])"; done
