"""Debug: a fuzz scene, fast vs counting variant, where do they differ (run on the GPU box)."""
import os, sys, random, ctypes
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
import test_gpu_fuzz as tf
import test_gpu_parity as tp
import pathlib, tempfile, math
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = pathlib.Path(tempfile.mkdtemp())
def torus(u, v):
    a, b = 2 * math.pi * u, 2 * math.pi * v
    return ((2 + 0.7 * math.cos(b)) * math.cos(a), (2 + 0.7 * math.cos(b)) * math.sin(a), 0.7 * math.sin(b))
def blob(u, v):
    a, b = 2 * math.pi * u, math.pi * (v - 0.5)
    r = 1.5 + 0.3 * math.sin(5 * a) * math.cos(3 * b)
    return (r * math.cos(b) * math.cos(a), r * math.cos(b) * math.sin(a), r * math.sin(b))
tp._write_uv_mesh(d / "torus.obj", 24, 10, torus)
tp._write_uv_mesh(d / "blob.obj", 20, 10, blob)
rnd = random.Random(1)
(d / "noise.ppm").write_bytes(b"P6\n16 16\n255\n" + bytes(rnd.randrange(256) for _ in range(16 * 16 * 3)))
rnd = random.Random(1000 + seed)
xml = d / "s.xml"
xml.write_text(tf._scene_xml(rnd, d, seed % 3 == 2))
print(xml.read_text())
scene = pkg.Scene.from_xml(str(xml))
W, H = 96, 64
ctx = pkg.Context(0)
ctx.upload(scene)
print("lists", ctx.light_lists())
frs = pkg.frame_setup(scene.desc.camera, W, H, collect_stats=True)
cnt, gst = ctx.render(frs, stats=True)
for flags in (0, 64):
    pkg.hip.rtu_debug_flags(ctx._h, flags)
    for thr in (10 ** 9, 1):
        fr = pkg.frame_setup(scene.desc.camera, W, H)
        fr.coop_threshold = thr
        fast, _ = ctx.render(fr)
        bad = np.argwhere((fast.view(np.uint32) != cnt.view(np.uint32)).any(axis=2))
        print("flags", flags, "thr", thr, "differing pixels", len(bad), bad[:10].tolist())
        for y, x in bad[:5]:
            print("   ", y, x, fast[y, x], cnt[y, x])
