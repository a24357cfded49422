#!/usr/bin/env python3
"""GPU-clock timeline of one frame (rtu_render_timeline): in-kernel stamps, no profiler.
usage: python tools/gpu_timeline.py [tag ...]   (tags under tests/golden/)"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as g

pkg = g.load_package()
ctx = pkg.Context(0)
for tag in sys.argv[1:] or ["teapot2_1080"]:
    gdir = os.path.join(REPO, "tests", "golden", tag)
    meta = json.load(open(os.path.join(gdir, "meta.json")))
    W, H = meta["width"], meta["height"]
    scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
    ctx.upload(scene)
    fr = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=0, shard_count=int(os.environ.get("TL_SHARDS", "1")))
    d = pkg.hip.rtu_device_alloc(ctx._h, W * H * 16)
    for _ in range(3):
        rows = ctx.render_timeline(fr, d)
    frames, deferred = ctx.frame_counts()
    print("%s  frames per level %s  deferred rays per phase (primary, L0..L5) %s" % (tag, frames, deferred))
    ms = ctx.time_render(fr, d, None, 20)
    print("%s  %dx%d  (unstamped: %.1f us/frame)" % (tag, W, H, ms * 1e3))
    prev_end = 0.0
    for name, t0, t1 in rows:
        print("  %-18s start %8.1f  end %8.1f  dur %7.1f  gap %6.1f" % (name, t0, t1, t1 - t0, t0 - prev_end))
        prev_end = max(prev_end, t1)
    print("  frame span %.1f us" % prev_end)
    if os.environ.get("TL_EXITS"):
        import numpy as np
        for name, t0, t1 in rows:
            if t1 - t0 < 10:
                continue
            e = ctx.timeline_exits(name)
            if len(e):
                q = np.percentile(e, [10, 50, 90, 99])
                print("    %-16s wavefront exits: n=%d p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (name, len(e), q[0], q[1], q[2], q[3], e.max()))
    pkg.hip.rtu_device_free(ctx._h, d)
ctx.close()
