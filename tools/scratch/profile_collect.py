#!/usr/bin/env python3
"""Turn gpurun_out/round/ (tools/profile_round.sh) into the committed evidence under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --stats summary of the default bench.py run
  <tag>_frame_timeline.txt kernel-by-kernel timeline of one frame
  <tag>_pmc_per_kernel.txt mean PMC counters per kernel and recursion level
  hbm_traffic.json         HBM bytes per frame (FETCH_SIZE + WRITE_SIZE summed over the
                           frame's kernels), read by bench.py into roofline.traffic
"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
def newest(pattern):
    """gpurun merges a run's files into what is already under gpurun_out/: take the most recent match."""
    return max(glob.glob(pattern), key=os.path.getmtime)


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = "gpurun_out/round"
os.makedirs("profiles", exist_ok=True)
st = newest(root + "/stats/*/*_kernel_stats.csv")
shutil.copyfile(st, "profiles/%s_kernel_stats.csv" % tag)
tl = subprocess.run([sys.executable, "tools/frame_timeline.py", newest(root + "/stats/*/*_kernel_trace.csv")], capture_output=True, text=True).stdout
open("profiles/%s_frame_timeline.txt" % tag, "w").write(tl)
pm = subprocess.run([sys.executable, "tools/pmc_summary.py", root + "/pmc_*/*/*_counter_collection.csv"], capture_output=True, text=True).stdout
open("profiles/%s_pmc_per_kernel.txt" % tag, "w").write(pm)

def per_frame(counter, d, root=root):
    """Mean counter sum per launch sequence (k_primary .. last k_combine) of the fast kernel variant."""
    f = newest(root + "/%s/*/*_counter_collection.csv" % d)
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "::k_" not in k or re.search(r"<[^>]*true", k) or r["Counter_Name"] != counter:
            continue
        per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = k
    frames, cur = [], None
    for d_ in sorted(per):
        if "k_primary<" in names[d_]:
            cur = 0.0
            frames.append(cur)
        if cur is not None:
            frames[-1] += per[d_]
    frames = frames[1:-1] if len(frames) > 4 else frames  # drop possibly partial first/last
    return sum(frames) / len(frames)

fetch_kb = per_frame("FETCH_SIZE", "pmc_fetch")
write_kb = per_frame("WRITE_SIZE", "pmc_write")
def fif_of(root):
    """Frames in flight of the run under `root`, from the bench line in its log."""
    for line in open(root + "/stats.log"):
        if line.startswith("{"):
            return json.loads(line)["config"]["frames_in_flight"]
    raise SystemExit("no bench line in " + root + "/stats.log")

by_fif = {str(fif_of(root)): int((fetch_kb + write_kb) * 1024)}
for extra in sorted(glob.glob("gpurun_out/round_fif*")):
    by_fif[str(fif_of(extra))] = int((per_frame("FETCH_SIZE", "pmc_fetch", extra) + per_frame("WRITE_SIZE", "pmc_write", extra)) * 1024)
    st2 = newest(extra + "/stats/*/*_kernel_stats.csv")
    shutil.copyfile(st2, "profiles/%s_%s_kernel_stats.csv" % (tag, os.path.basename(extra).replace("round_", "")))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 48 --warmup 16 --no-cpu [--frames-in-flight n]",
       "fetch_size_kb_per_launch_sequence": fetch_kb, "write_size_kb_per_launch_sequence": write_kb,
       "bytes_per_launch_by_frames_in_flight": by_fif,
       "note": "FETCH_SIZE is uncalibrated for narrow / gather accesses on gfx950 (it halves wide streaming reads); "
               "reported as counted, not doubled, because this path has no wide streaming read",
       "bytes_per_launch": int((fetch_kb + write_kb) * 1024)}
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out))
