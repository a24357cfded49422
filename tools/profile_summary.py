#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/profile_bench.sh) -> the evidence committed under profiles/:
  <tag>_kernel_stats.csv    rocprofv3 --stats summary of the bench command, as written by rocprofv3
  <tag>_per_kernel.txt      per kernel SLOT of the timed launch sequences (k_primary, k_trace2(L0), ...: the names bench.py and
                            rtu_kernel_slot_name use): mean duration, HBM bytes, VALU instructions per wave, lane utilisation, waits
  <tag>_hbm_traffic.json    HBM bytes per launch per slot (FETCH_SIZE + WRITE_SIZE)
  <tag>_per_kernel.json     the same per slot as numbers — us, hbm_bytes_per_launch, valu_issue_frac (VALU wave-instructions per second against
                            CUs x 4 SIMDs x clock / 4: a wave64 VALU instruction occupies a 16-lane SIMD for four cycles), lanes_active, waiting_frac —
                            read by bench.py into roofline.{traffic, hbm_counter_frac, valu_issue_frac, lanes_active, bound}
Only dispatches of the timed feature set (kernel template argument = the bench's frames-in-flight set, 4 / 5, or 0 / 1 with
--frames-in-flight 1) are used; the levels of k_trace / k_trace2 / k_consume / k_combine are told apart by their order inside
a launch sequence (a sequence starts at k_node_rects / k_primary)."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = "gpurun_out/" + tag
feat = sys.argv[2] if len(sys.argv) > 2 else "4"


def newest(pattern):
    m = glob.glob(pattern)
    if not m:
        raise SystemExit("nothing matches " + pattern)
    return max(m, key=os.path.getmtime)


def slot_names(rows):
    """rows: dispatches in time order as (id, kernel name). Yields (id, slot name) for the kernels of feature set `feat`."""
    out = {}
    seq = collections.Counter()
    levels = None
    for did, name in rows:
        m = re.search(r"::(k_\w+)<([^>]*)>", name)
        if not m:
            if "k_node_rects" in name or "k_prelude" in name:
                seq.clear()
            continue
        k, targs = m.group(1), [t.strip() for t in m.group(2).split(",")]
        if targs[-1] != feat or "true" in targs or k.endswith("_counting"):
            continue
        if k == "k_primary":
            seq.clear()
        if k in ("k_primary", "k_primary2", "k_primary2c"):
            out[did] = k
        elif k == "k_combine":
            out[did] = ("k_combine", seq[k])  # launched from the deepest level down: renumbered below
            seq[k] += 1
        elif k == "k_tail":
            out[did] = "k_tail"
        elif k == "k_trace":
            out[did] = "k_trace(L%d)" % seq[k]
            seq[k] += 1
        else:
            # the stage-2 kernels and k_consume of a level follow its k_trace (an idle stage-2 launch may have been dropped: the level is
            # the last k_trace's, not the count of this kernel's own launches)
            out[did] = "%s(L%d)" % (k, max(seq["k_trace"] - 1, 0))
    # k_combine runs bottom-up: the LAST one of a sequence is level 0
    ids = sorted(out)
    i = 0
    while i < len(ids):
        if isinstance(out[ids[i]], tuple):
            j = i
            while j < len(ids) and isinstance(out[ids[j]], tuple):
                j += 1
            n = j - i
            for t in range(i, j):
                out[ids[t]] = "k_combine(L%d)" % (n - 1 - (t - i))
            i = j
        else:
            i += 1
    return out


def trace_rows(f):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


os.makedirs("profiles", exist_ok=True)
shutil.copyfile(newest(root + "/stats/*/*_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % tag)
tr = trace_rows(newest(root + "/stats/*/*_kernel_trace.csv"))
names = slot_names([(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in tr])
dur = collections.defaultdict(list)
for r in tr:
    s = names.get(int(r["Dispatch_Id"]))
    if s:
        dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(root + "/pmc_*")):
    try:
        f = newest(d + "/*/*_counter_collection.csv")
        t = trace_rows(newest(d + "/*/*_kernel_trace.csv"))
    except SystemExit:
        continue
    nm = slot_names([(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in t])
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        did = int(r["Dispatch_Id"])
        if did in nm:
            per[(did, r["Counter_Name"])] += float(r["Counter_Value"])
    for (did, c), v in per.items():
        counters[nm[did]][c].append(v)


def mean(v):
    return sum(v) / len(v) if v else float("nan")


lines = []
traffic = {}
perk = {}
CUS, CLOCK_GHZ = 256, 2.4   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
VALU_PEAK = CUS * 4 * CLOCK_GHZ * 1e9 / 4  # wave64 VALU instructions per second
order = sorted(dur, key=lambda k: -mean(dur[k]))
total = sum(mean(dur[k]) for k in order)
lines.append("# %s: kernel slots of the timed launch sequences (feature set %s), mean over %d sequences; rocprofv3 kernel trace + PMC passes" % (tag, feat, len(dur.get("k_primary", []))))
lines.append("# sum of the mean kernel durations: %.1f us" % total)
for k in order:
    c = {n: mean(v) for n, v in counters[k].items()}
    waves = c.get("SQ_WAVES", float("nan"))
    l = "%-16s %8.1f us (%4.1f %%)" % (k, mean(dur[k]), 100 * mean(dur[k]) / total)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        fb, wb = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
        traffic[k] = {"fetch_bytes": int(fb), "write_bytes": int(wb), "hbm_bytes_per_launch": int(fb + wb), "us": round(mean(dur[k]), 2)}
        l += "  HBM fetch %7.1f MB write %7.1f MB -> %6.1f GB/s" % (fb / 1e6, wb / 1e6, (fb + wb) / mean(dur[k]) / 1e3)
    if "SQ_INSTS_VALU" in c:
        l += "  waves %8d  VALU/wave %6.0f SALU/wave %5.0f VMEM_RD/wave %5.1f" % (waves, c["SQ_INSTS_VALU"] / waves, c["SQ_INSTS_SALU"] / waves, c["SQ_INSTS_VMEM_RD"] / waves)
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        # lanes active per VALU instruction: SQ_THREAD_CYCLES_VALU / (64 lanes x SQ_ACTIVE_INST_VALU), as in round 1
        l += "  lanes active %4.0f %%" % (100 * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]))
        if "SQ_BUSY_CYCLES" in counters[k]:
            pass
    if "SQ_WAIT_ANY" in c and "SQ_ACTIVE_INST_ANY" in c:
        tot = c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]
        l += "  wave-cycles: waiting %2.0f %% issue-stalled %2.0f %% issuing %2.0f %%" % (100 * c["SQ_WAIT_ANY"] / tot, 100 * c["SQ_WAIT_INST_ANY"] / tot, 100 * c["SQ_ACTIVE_INST_ANY"] / tot)
    if "TCC_HIT_sum" in c:
        l += "  L2 hit %3.0f %%" % (100 * c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))
    lines.append(l)
    e = {"us": round(mean(dur[k]), 2)}
    if k in traffic:
        e["hbm_bytes_per_launch"] = traffic[k]["hbm_bytes_per_launch"]
    if "SQ_INSTS_VALU" in c:
        e["waves"] = int(waves)
        e["valu_per_wave"] = round(c["SQ_INSTS_VALU"] / waves, 1)
        e["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] / (mean(dur[k]) * 1e-6) / VALU_PEAK, 4)
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_active"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]), 4)
    if "SQ_WAIT_ANY" in c and "SQ_ACTIVE_INST_ANY" in c:
        e["waiting_frac"] = round(c["SQ_WAIT_ANY"] / (c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]), 4)
    if "TCC_HIT_sum" in c:
        e["l2_hit"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    perk[k] = e
open("profiles/%s_per_kernel.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if traffic:
    fif = None
    for line in open(root + "/stats.log"):
        if line.startswith("{"):
            fif = json.loads(line)["config"]["frames_in_flight"]
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the command of tools/profile_bench.sh (gpurun_out/<tag>/command.txt)",
           "commit": commit, "frames_in_flight": fif,
           "note": "FETCH_SIZE / WRITE_SIZE as counted (KB x 1024). MI355X_MICROARCH.md: FETCH_SIZE halves wide coalesced 16-B-per-lane streaming reads and is "
                   "uncalibrated for other widths; this path's reads are per-lane gathers and 16-byte record reads, so the fetch figure is a lower bound "
                   "(at most 2x low); writes are counted exactly",
           "per_kernel": traffic,
           "hbm_bytes_per_launch_sequence": int(sum(v["hbm_bytes_per_launch"] for v in traffic.values()))}
    json.dump(out, open("profiles/%s_hbm_traffic.json" % tag, "w"), indent=1)
    cmd = open(root + "/command.txt").read().strip() if os.path.exists(root + "/command.txt") else "python3 bench.py"
    json.dump({"source": "rocprofv3 --kernel-trace --stats, then separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ instruction counts; SQ activity / wait / TCC hit) over the command below",
               "command": cmd, "commit": commit, "frames_in_flight": fif, "valu_peak_wave_instructions_per_s": VALU_PEAK,
               "sum_of_mean_kernel_us": round(total, 1), "per_kernel": perk}, open("profiles/%s_per_kernel.json" % tag, "w"), indent=1)
