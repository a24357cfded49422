#!/usr/bin/env python3
"""Per deferred primary ray the work units of its BVH walk (2 per inner step + 1 per triangle test): rtu_debug_flags 131072 in touched-bytes
mode writes them over the red channel (include/rtu_render.h). Prints the distribution; the image goes to gpurun_out/walk_units.npy."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else "teapot2_1080"
gdir = os.path.join(REPO, "tests", "golden", tag)
meta = json.load(open(os.path.join(gdir, "meta.json")))
W, H = meta["width"], meta["height"]
scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
ctx = pkg.Context(0)
ctx.upload(scene)
n = 4
cams = [scene.desc.camera] * n
d = pkg.hip.rtu_device_alloc(ctx._h, n * W * H * 16)
pkg.hip.rtu_debug_flags(ctx._h, 131072 | 8192)
fs = [pkg.frame_setup(c, W, H, collect_stats=2) for c in cams]
base = np.empty((n, H, W, 4), np.float32)
for rep in range(3):
    while True:
        ctx.render_frames_device(fs, d, None)
        try:
            ctx.frame_status(); break
        except pkg.RtuError as e:
            if e.code != pkg.RTU_ERR_CAPACITY: raise
pkg.hip.rtu_copy_to_host(ctx._h, base.ctypes.data, d, base.nbytes)
pkg.hip.rtu_debug_flags(ctx._h, 8192)
ref = np.empty((n, H, W, 4), np.float32)
ctx.render_frames_device(fs, d, None); ctx.frame_status()
pkg.hip.rtu_copy_to_host(ctx._h, ref.ctypes.data, d, ref.nbytes)
diff = base[0, ..., 0] != ref[0, ..., 0]
u = base[0, ..., 0][diff]
print("pixels whose red channel was replaced:", int(diff.sum()), "(deferred pixels; a few coincide by value)")
print("units: mean %.1f median %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % (u.mean(), np.median(u), *np.percentile(u, [90, 99, 99.9]), u.max()))
for thr in (64, 128, 256, 512):
    print("  > %d units: %d rays (%.2f %%), %.1f %% of all units" % (thr, int((u > thr).sum()), 100.0 * (u > thr).mean(), 100.0 * u[u > thr].sum() / u.sum()))
ys, xs = np.nonzero(diff & (base[0, ..., 0] > 64))
hit = ref[0, ..., 3] < 1e29
if len(xs):
    print("rays beyond 64 units: bounding box x %d..%d y %d..%d; %.0f %% of them hit something nearer than the background" % (xs.min(), xs.max(), ys.min(), ys.max(), 100.0 * hit[ys, xs].mean()))
# a wavefront of 64 neighbouring deferred rays lasts as long as its longest walk
v = np.sort(u)
rng = np.random.default_rng(1)
idx = np.flatnonzero(diff.ravel())
vals = base[0, ..., 0].ravel()[idx]
groups = vals[: len(vals) // 64 * 64].reshape(-1, 64)  # (image order: neighbours, as the defer lists hold them)
print("wavefronts of 64 rays in image order: mean of the maxima %.1f units, mean of the means %.1f -> lanes busy %.0f %%" % (groups.max(1).mean(), groups.mean(1).mean(), 100.0 * groups.mean(1).mean() / groups.max(1).mean()))
np.save(os.path.join(REPO, "gpurun_out", "walk_units.npy"), np.where(diff, base[0, ..., 0], 0).astype(np.uint16))
