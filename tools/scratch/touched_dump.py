#!/usr/bin/env python3
"""Per-kernel touched counters (collect_stats = 2) of one launch sequence: tools/scratch/touched_dump.py [tag] [frames]"""
import json, os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import __graft_entry__ as g
pkg = g.load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else "teapot2_1080"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gdir = os.path.join(REPO, "tests", "golden", tag)
meta = json.load(open(os.path.join(gdir, "meta.json")))
W, H = meta["width"], meta["height"]
scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
ctx = pkg.Context(0)
ctx.upload(scene)
d = pkg.hip.rtu_device_alloc(ctx._h, W * H * 16 * nb)
frs = []
for j in range(nb):
    fr = pkg.frame_setup(scene.desc.camera, W, H)
    fr.collect_stats = 2
    frs.append(fr)
for _ in range(3):
    while True:
        try:
            if nb == 1: ctx.render_device(frs[0], d, None)
            else: ctx.render_frames_device(frs, d, None)
            ctx.frame_status()
            break
        except pkg.RtuError as e:
            if e.code == pkg.RTU_ERR_CAPACITY: continue
            raise
t = ctx.touched()
for k, v in t.items():
    print(k, {a: b for a, b in v.items() if b})
