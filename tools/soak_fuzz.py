#!/usr/bin/env python3
"""Extended soak of the fast variant against the counting variant on the random scenes of tests/test_gpu_fuzz.py (more seeds than
the test suite affords): both stage-2 forms, a ragged resolution, three shards, a batch of three turned cameras. Prints the seeds
that differ. usage: python tools/soak_fuzz.py FIRST_SEED LAST_SEED"""
import math, os, random, sys, tempfile, pathlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import __graft_entry__ as g
import importlib.util
pkg = g.load_package()
spec = importlib.util.spec_from_file_location("fz", os.path.join(REPO, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
d = pathlib.Path(tempfile.mkdtemp())


def torus(u, v):
    a, b = 2 * math.pi * u, 2 * math.pi * v
    return ((2 + 0.7 * math.cos(b)) * math.cos(a), (2 + 0.7 * math.cos(b)) * math.sin(a), 0.7 * math.sin(b))


def blob(u, v):
    a, b = 2 * math.pi * u, math.pi * (v - 0.5)
    r = 1.5 + 0.3 * math.sin(5 * a) * math.cos(3 * b)
    return (r * math.cos(b) * math.cos(a), r * math.cos(b) * math.sin(a), r * math.sin(b))


fz._write_uv_mesh(d / "torus.obj", 24, 10, torus)
fz._write_uv_mesh(d / "blob.obj", 20, 10, blob)
rnd0 = random.Random(1)
(d / "noise.ppm").write_bytes(b"P6\n16 16\n255\n" + bytes(rnd0.randrange(256) for _ in range(16 * 16 * 3)))
first, last = int(sys.argv[1]), int(sys.argv[2])
only = [int(v) for v in os.environ.get("SOAK_SEEDS", "").split(",") if v]
dbg = int(os.environ.get("SOAK_DBG", "0"))
spp = int(os.environ.get("SOAK_SAMPLES", "0"))  # > 0: recipe S with that many samples per pixel (glossy bounces, soft shadows, lens); single frames and shards only
ctx = pkg.Context(0)
bad = []
W, H = 157, 99
def same(a, b, what, seed):
    eq = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if not eq:
        diff = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1))
        zdiff = int((a[..., 3].view(np.uint32) != b[..., 3].view(np.uint32)).sum())
        print("  seed %d: %s differs at %d pixels (%d in z), first %s: %s vs %s" % (seed, what, len(diff), zdiff, diff[0], a[tuple(diff[0])], b[tuple(diff[0])]), flush=True)
    return eq


for seed in (only or range(first, last)):
    rnd = random.Random(1000 + seed)
    xml = d / ("s%d.xml" % seed)
    xml.write_text(fz._make_stochastic(fz._scene_xml(rnd, d, seed % 3 == 2), rnd) if spp else fz._scene_xml(rnd, d, seed % 3 == 2))
    scene = pkg.Scene.from_xml(str(xml))
    ctx.upload(scene)
    if dbg:
        pkg.hip.rtu_debug_flags(ctx._h, dbg)
    cams = []
    for i in range(3):
        cam = type(scene.desc.camera).from_buffer_copy(scene.desc.camera)
        cam.pos[0] += 1.1 * i
        cam.fov += 7.0 * i
        cams.append(cam)
    kw = {"samples": spp} if spp else {}
    refs = [ctx.render(pkg.frame_setup(c, W, H, collect_stats=True, **kw), stats=True)[0] for c in (cams[:1] if spp else cams)]
    ok = True
    for thr in (10 ** 9, 1):
        fr = pkg.frame_setup(cams[0], W, H, **kw)
        fr.coop_threshold = thr
        ok &= same(ctx.render(fr)[0], refs[0], "single frame thr %d" % thr, seed)
        shards, frames = [], []
        for r in range(3):
            f = pkg.frame_setup(cams[0], W, H, shard_rank=r, shard_count=3, **kw)
            f.coop_threshold = thr
            shards.append(ctx.render(f)[0])
            frames.append(f)
        ok &= same(pkg.assemble(shards, frames, H), refs[0], "3 shards thr %d" % thr, seed)
        if spp:
            continue  # (frames in flight are recipe W)
        fb = [pkg.frame_setup(c, W, H) for c in cams]
        for f in fb:
            f.coop_threshold = thr
        dptr = pkg.hip.rtu_device_alloc(ctx._h, 3 * W * H * 16)
        for rep in range(3):  # (the later launches of the shape follow the habits the first taught the context: side mode, k_tail, grid hints)
            for attempt in range(17):  # the asynchronous entry's contract: render again until no level ran out of capacity (a level per round at worst)
                ctx.render_frames_device(fb, dptr, None)
                try:
                    ctx.frame_status()
                    break
                except pkg.RtuError as e:
                    if e.code != pkg.RTU_ERR_CAPACITY or attempt == 16:
                        raise
            out = np.empty((3, H, W, 4), np.float32)
            pkg.hip.rtu_copy_to_host(ctx._h, out.ctypes.data, dptr, out.nbytes)
            for i in range(3):
                ok &= same(out[i], refs[i], "batch frame %d thr %d launch %d" % (i, thr, rep), seed)
        pkg.hip.rtu_device_free(ctx._h, dptr)
    if not ok:
        bad.append(seed)
        print("seed %d DIFFERS" % seed, flush=True)
    if seed % 25 == 0:
        print("... seed %d" % seed, flush=True)
print("seeds %d..%d: %d differ %s" % (first, last - 1, len(bad), bad))
ctx.close()
