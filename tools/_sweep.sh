for thr in 100000 200000 400000 800000 2000000; do for t in teapot2_1080 p11_1080; do
timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --tag $t --coop-threshold $thr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"thr $thr $t\", d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], d[\"config\"][\"z_bit_exact_vs_reference_golden\"])"
done; done
