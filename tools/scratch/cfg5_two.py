#!/usr/bin/env python3
"""config 5 (Project11 @1080p, recipe P) as C contexts on ONE GPU, each rendering every C-th band from a host thread of its own:
does the overlap of two or three launch sequences pay for recipe P as it does for recipe W?  usage: cfg5_two.py [spp] [C ...]"""
import json, os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import __graft_entry__ as g
pkg = g.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Cs = [int(v) for v in sys.argv[2:]] or [1, 2, 3]
gdir = os.path.join(REPO, "tests", "golden", "p11_1080")
meta = json.load(open(os.path.join(gdir, "meta.json")))
W, H = meta["width"], meta["height"]
scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
for C in Cs:
    ctxs = [pkg.Context(0) for _ in range(C)]
    bufs, frames = [], []
    for r, c in enumerate(ctxs):
        c.upload(scene)
        fr = pkg.frame_setup(scene.desc.camera, W, H, shard_rank=r, shard_count=C, samples=spp)
        fr.gather_bounces = 4
        frames.append(fr)
        bufs.append(pkg.hip.rtu_device_alloc(c._h, pkg.shard_rows(fr) * W * 16))
    def work(i):
        ctxs[i].render_device(frames[i], bufs[i], None)
        ctxs[i].frame_status()
    def once():
        ts = [threading.Thread(target=work, args=(i,)) for i in range(C)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        return time.perf_counter() - t0
    once()
    best = min(once() for _ in range(3))
    print("contexts %d: %.2f ms per %d-spp frame" % (C, best * 1e3, spp), flush=True)
    for c, b in zip(ctxs, bufs):
        pkg.hip.rtu_device_free(c._h, b)
        c.close()
