timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for fl in 0 2; do for thr in 100000 150000 200000; do for t in teapot2_1080 p11_1080; do
timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --tag $t --frame-flags $fl --coop-threshold $thr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"flags $fl thr $thr $t\", d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], d[\"config\"][\"z_bit_exact_vs_reference_golden\"])"
done; done; done
