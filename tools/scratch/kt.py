#!/usr/bin/env python3
"""Print the per-kernel table of a bench.py JSON line (file argument or stdin)."""
import json, sys
d = json.loads([l for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin) if l.startswith("{")][-1])
c = d["config"]
print("value %.0f Mrays/s  ms/frame %.4f  batch %.4f ms  single %s  z_ok %s" % (d["value"], d["ms_per_step"], c["frame_latency_ms"], (c.get("single_frame") or {}).get("ms_per_frame"), c["z_bit_exact_vs_reference_golden"]))
r = d["roofline"]
print("dominant %s %.4f ms  achieved %.0f GB/s frac %.3f l2 %.3f" % (r.get("kernel"), r.get("kernel_ms") or 0, r.get("achieved") or 0, r.get("frac") or 0, r.get("l2_frac") or 0))
for k, v in (r.get("kernels") or {}).items():
    if v["ms"] >= 0.01:
        print("  %-16s %7.1f us  %8.1f MB  rays %9d" % (k, v["ms"] * 1e3, v["bytes"] / 1e6, v["rays"]))
