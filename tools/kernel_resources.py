#!/usr/bin/env python3
"""Register / spill table of the kernels of one feature set, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage on csrc/render_feat<F>.hip, the flags of raytracer-utah_amd/Makefile).
usage: tools/kernel_resources.py [feature set = 4] [stack = 32] > profiles/rNN_kernel_resources.txt
Only the instantiations with BVH stack size `stack` are listed (the sizes differ in the LDS stack only)."""
import os, re, subprocess, sys

feat = sys.argv[1] if len(sys.argv) > 1 else "4"
stack = sys.argv[2] if len(sys.argv) > 2 else "32"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(repo, "raytracer-utah_amd")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-I" + os.path.join(repo, "include"),
       "-I" + os.path.join(pkg, "csrc"), "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null",
       os.path.join(pkg, "csrc", "render_feat%s.hip" % feat)] + os.environ.get("RTU_EXTRA", "").split()
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
name = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
print("# %s" % " ".join(cmd[1:-1] + [os.path.relpath(cmd[-1], repo)] if not os.environ.get("RTU_EXTRA") else cmd[1:]))
print("%-46s %5s %5s %7s %7s %8s %5s %6s" % ("kernel", "SGPR", "VGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
for r in rows:
    n = name(r["name"]).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(KernelArgs.*", "", n).replace("void ", "")
    m = re.search(r"<(\d+), ", n)
    if m and m.group(1) != stack:
        continue
    print("%-46s %5s %5s %7s %7s %8s %5s %6s" % (n, r.get("TotalSGPRs", "?"), r.get("VGPRs", "?"), r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"),
                                                   r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
