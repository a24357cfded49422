#!/usr/bin/env python3
"""Mean duration per kernel, in launch order, of the launch sequences of one feature set in a rocprofv3 --kernel-trace CSV
(feature set = last template argument of the kernels: 4 = frames in flight, 0 = one frame; k_prelude starts a sequence).
usage: tools/seq_timeline.py <kernel_trace.csv glob> [feature set = 4]"""
import collections, csv, glob, re, sys
pat = sys.argv[1]
feat = sys.argv[2] if len(sys.argv) > 2 else "4"
for f in sorted(glob.glob(pat)):
    rows = [r for r in csv.DictReader(open(f)) if "::k_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seqs, cur = [], None
    for r in rows:
        n = r["Kernel_Name"].split("::")[-1].split("(")[0]
        m = re.search(r"<([^>]*)>", n)
        targs = [t.strip() for t in m.group(1).split(",")] if m else []
        if n.startswith("k_prelude"):
            cur = []
            seqs.append(cur)
            continue
        if cur is None or n.startswith("k_tile_occ"):
            continue
        if targs and targs[-1] != feat:
            cur = None if not cur else cur
            if cur is not None and not cur:
                seqs.pop()
                cur = None
            continue
        cur.append((re.sub(r"<.*", "", n), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    seqs = [s for s in seqs if s]
    if not seqs:
        continue
    n0 = max(collections.Counter(len(s) for s in seqs).items(), key=lambda kv: kv[1])[0]
    seqs = [s for s in seqs if len(s) == n0][2:]  # the first ones still learn their grids
    print("%s: %d sequences of %d kernels (feature set %s)" % (f, len(seqs), n0, feat))
    tot = 0.0
    for i in range(n0):
        d = sum(s[i][1] for s in seqs) / len(seqs)
        tot += d
        print("  %2d %-14s %8.1f us" % (i, seqs[0][i][0], d))
    span = sum((s[-1][3] - s[0][2]) / 1e3 for s in seqs) / len(seqs)
    print("  sum %.1f us, span first start -> last end %.1f us" % (tot, span))
