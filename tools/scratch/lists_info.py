import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
for tag in sys.argv[1:] or ["teapot2_1080", "p11_1080"]:
    scene = pkg.Scene.from_blob_file(os.path.join("tests", "golden", tag, "scene.rtus.gz"))
    ctx = pkg.Context(0)
    ctx.upload(scene)
    W, H = scene.desc.camera.img_width, scene.desc.camera.img_height
    print(tag, W, H, ctx.light_lists())
    fr = pkg.frame_setup(scene.desc.camera, 1920, 1080)
    ctx.render(fr)
    print("  frames, deferred", ctx.frame_counts())
    ctx.close()
