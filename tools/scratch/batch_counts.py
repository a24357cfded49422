import os, sys, importlib.util
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
spec = importlib.util.spec_from_file_location("b", os.path.join(os.path.dirname(__file__), "..", "..", "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
scene = pkg.Scene.from_blob_file("tests/golden/teapot2_1080/scene.rtus.gz")
ctx = pkg.Context(0); ctx.upload(scene)
for B in (1, 20, 32):
    cams = [b.orbit_camera(scene.desc.camera, 2.0 * j) for j in range(B)]
    frames = [pkg.frame_setup(c, 1920, 1080) for c in cams]
    d = pkg.hip.rtu_device_alloc(ctx._h, B * 1920 * 1080 * 16)
    for _ in range(4):
        if B == 1: ctx.render_device(frames[0], d, None)
        else: ctx.render_frames_device(frames, d, None)
        try: ctx.frame_status()
        except pkg.RtuError as e: print("retry", e)
    print(B, ctx.frame_counts())
    pkg.hip.rtu_device_free(ctx._h, d)
