"""Experiment: K contexts on one GPU, each on its own stream, each with B frames in flight, launched alternately —
do the latency-bound deep levels of one launch sequence overlap the wide kernels of another?"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
tag = "teapot2_1080"
gdir = os.path.join("tests", "golden", tag)
meta = json.load(open(os.path.join(gdir, "meta.json")))
W, H = meta["width"], meta["height"]
scene = pkg.Scene.from_blob_file(os.path.join(gdir, "scene.rtus.gz"))
rays = 2558987
for K, B in [(1, 32), (2, 16), (2, 32), (4, 8), (4, 16), (3, 16)]:
    ctxs = [pkg.Context(0) for _ in range(K)]
    for c in ctxs:
        c.upload(scene)
    streams = [torch.cuda.Stream() for _ in range(K)]
    bufs = [torch.zeros(B * W * H * 4, dtype=torch.float32, device="cuda") for _ in range(K)]
    cams = [bench.orbit_camera(scene.desc.camera, 2.0 * j) for j in range(B)]
    frames = [pkg.frame_setup(c, W, H) for c in cams]
    def run(n):
        for i in range(n):
            k = i % K
            ctxs[k].render_frames_device(frames, bufs[k].data_ptr(), streams[k].cuda_stream)
    for attempt in range(4):
        run(2 * K)
        torch.cuda.synchronize()
        ok = True
        for c in ctxs:
            try:
                c.frame_status()
            except pkg.RtuError:
                ok = False
        if ok:
            break
    n = 24 * K
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for c in ctxs:
        c.frame_status()
    print("contexts %d x %2d frames in flight: %.4f ms/frame  %.1f Grays/s" % (K, B, el / (n * B) * 1e3, rays * n * B / el / 1e9), flush=True)
    for c in ctxs:
        c.close()
    del bufs
    torch.cuda.empty_cache()
