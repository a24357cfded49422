#!/usr/bin/env python3
"""Loops of one kernel in hipcc -S output: per backward branch the body's instruction mix (VALU / SALU / memory / spill moves).
usage: isa_loops.py file.s kernel_substring"""
import re, sys
src = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
end = next(i for i in range(start + 1, len(src)) if src[i].startswith("\t.end_amdhsa_kernel") or src[i].startswith(".Lfunc_end"))
body = src[start:end]
labels = {}
ins = []
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = len(ins)
    elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        ins.append(l.strip())
print("kernel", body[0].split(":")[0], "instructions", len(ins))
def kind(s):
    op = s.split()[0]
    if op.startswith(("v_readlane", "v_writelane")): return "lane"
    if op.startswith(("scratch_", "buffer_")) : return "scratch"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_load", "s_buffer")): return "smem"
    if op.startswith(("global_", "flat_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
loops = []
for i, s in enumerate(ins):
    m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", s)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i))
tot = {}
for s in ins: tot[kind(s)] = tot.get(kind(s), 0) + 1
print("whole:", tot)
loops.sort(key=lambda ab: ab[1] - ab[0])
for a, b in loops:
    c = {}
    for s in ins[a:b + 1]: c[kind(s)] = c.get(kind(s), 0) + 1
    print("loop %6d..%6d len %5d " % (a, b, b - a + 1), " ".join("%s=%d" % kv for kv in sorted(c.items())))
