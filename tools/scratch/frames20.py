import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import math
import __graft_entry__ as g
pkg = g.load_package()
gdir = os.path.join(REPO, "tests", "golden", "teapot2_1080")
meta = json.load(open(os.path.join(gdir, "meta.json")))
W, H = meta["width"], meta["height"]
scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
ctx = pkg.Context(0); ctx.upload(scene)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import bench
cams = [bench.orbit_camera(scene.desc.camera, 2.0 * i) for i in range(n)]
fs = [pkg.frame_setup(c, W, H) for c in cams]
d = pkg.hip.rtu_device_alloc(ctx._h, n * W * H * 16)
for rep in range(4):
    while True:
        ctx.render_frames_device(fs, d, None)
        try:
            ctx.frame_status(); break
        except pkg.RtuError as e:
            if e.code != pkg.RTU_ERR_CAPACITY: raise
    print(rep, ctx.frame_counts())
